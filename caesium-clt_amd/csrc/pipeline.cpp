// pipeline.cpp -- the device batch queue: what caesium-clt's rayon par_iter over files
// (/root/reference/src/compressor.rs:74-101) becomes on an MI355X.  One csh_batch = one group of input
// files resident in HBM; csh_batch_run pushes the whole group through
//   entropy decode -> pixel-domain transcode -> masks/flags/runs -> stats/tables -> sizes/scan -> pack
//   -> stuffing/assembly
// on one stream with no host round trip (every size and offset is produced by device scans).
// Host work is limited to container logic: marker parsing, table/script setup, descriptor building.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <atomic>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/caesium_hip.h"
#include "devmem.hpp"
#include "jpeg_host.hpp"
#include "kernels.h"
#include "webp_kernels.h"
#include "../../include/vp8_tables.h"
#include "resize_host.h"

static thread_local char g_err[512];
void csh_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
extern "C" const char *csh_last_error(void) { return g_err; }

#ifdef CSH_EMUL
thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
int csh_emul_reverse = 0;
thread_local int csh_emul_phase = 0;
extern "C" void csh_emul_set_reverse(int r) { csh_emul_reverse = r; }
namespace csh { extern int csh_emul_jacobi; }
extern "C" void csh_emul_set_jacobi(int j) { csh::csh_emul_jacobi = j; }
#endif

namespace csh {

struct Item {
    int code = 0;
    std::string msg;
    JpegInfo in;
    JpegInfo out;     // output geometry (comp ids / sampling / tq)
    int image = -1;   // index among the images that reached the device, or -1
    size_t file_size = 0;
    std::vector<uint8_t> meta_out;  // APPn/COM segments that survive the metadata/ICC policy (frame header rebuilds)
};

}  // namespace csh

using namespace csh;

struct csh_batch {
    int device = 0;
    hipStream_t stream = 0;
    bool have_stream = false;
    CCSParameters params;
    std::vector<Item> items;
    int nimg = 0;
    bool lossless = false;
    bool rgb_out = false;          // csh_batch_create_pixels: stop after the resize branch's RGB
    bool webp = false;             // target container: the decoded (and resized) RGB goes to the VP8 encoder instead of the JPEG one
    uint32_t webp_mb_bytes = 768;  // output bytes reserved per macroblock (grows on overflow)
    int test_pool_shift = -1;      // CSH_TEST_POOL_SHIFT as read at the first pool layout of this batch (-1: not read yet)
    std::vector<csw::WebpImg> wimgs;
    uint64_t wwork_bytes = 0, wlevels = 0;
    uint32_t wmax_luma = 0;
    bool progressive = true;
    bool retain_dct = false;       // size targeting: keep the unquantised DCT so that another quality only re-quantises
    bool have_dct = false;
    int q_base = 0;                // quants[q_base + q] = output table for quality q (1..100)

    // host-side descriptor arrays
    std::vector<ImgDesc> imgs;
    std::vector<DecScan> dscans;
    std::vector<DevHuffSet> hsets;
    std::vector<ParHuffSet> phsets;   // the same sets in the parallel decoder's LDS form
    std::vector<char> phset_fits;     // 0: sub-table pool overflow -> sequential decoder
    std::vector<ParHuffSet4> phsets4; // compact form (types.h); slot4[set][0..3 DC, 4..7 AC] = slot or -1
    std::vector<std::array<int8_t, 8>> slot4;
    bool use4 = true;                 // every table set of the batch fits the compact form
    std::vector<DevQuant> quants;
    std::vector<PlaneWork> pwork;
    std::vector<ResizeWork> rwork;
    std::vector<ResizeTap> rtaps;
    std::vector<float> rweights;
    uint64_t rgb_bytes = 0, tmp_floats = 0, max_tmp = 0, max_dst = 0;
    uint32_t max_row_in = 0, max_out_w = 0, max_nh = 0;   // resize launches: samples per source row, pixels per resized row, resized rows
    uint32_t max_src_px = 0;
    std::vector<ParScan> pscans;
    std::vector<uint32_t> need_seq_init;
    std::vector<ProgChain> chains;        // progressive inputs (k_decode_prog.hip)
    std::vector<int> chain_scans;
    uint32_t refine_hist = 0, refine_pos = 0, refine_max_blocks = 0;   // AC refinement chains (k_decode_refine.hip): history masks, block positions, largest chain
    uint32_t total_sub = 0, max_sub = 0, max_par_blocks = 0, dc_total = 0;
    std::vector<EncScan> script;
    std::vector<ScanWork> swork;
    // per (work item, 256-unit chunk) slot: its work item, its SlotRec, its place in the list-coded / token-coded slot lists -- ~3.9 k slots per 1080p image
    // under the scan search (64 MB of records per 256 files): the host only counts them, k_make_slots writes them on the device from the work items
    uint32_t nslots = 0, nlist_slots = 0, ntok_slots = 0;
    uint64_t total_corr = 0;              // correction words: one per unit of a refinement scan
    std::vector<TokPlan> plans;
    std::vector<int> plan_comp, plan_image;
    std::vector<EChunk> echunks;          // the token kernel's workgroups
    uint64_t tok_cap = 0;                 // token pool capacity: the sum of the regions
    uint32_t tok_scale = 1;               // grows on overflow
    std::vector<TokRegion> regions;       // one per TokPlan, then one per DC / sequential work item
    std::vector<uint32_t> region_est;     // estimated tokens of each (x tok_scale = its capacity)
    uint32_t hist_rows = 0;               // rows of 256 symbol counts over all slots
    // the compacted coefficient lists the progressive AC first-pass scans are coded from (k_aclist.hip; types.h NzList)
    std::vector<NzList> nzlists;          // one per (image, component, Al) some scan of the batch needs
    std::vector<NzSet> nzsets;            // one per (image, component)
    std::vector<int> nzset_of;            // [image * CSH_MAX_COMPS + component] -> NzSet, -1
    std::vector<uint32_t> nzset_built;    // per set: levels some stage's builder makes
    std::vector<int> nzset_comp, nzset_image;
    std::vector<NzChunk> nzchunks;        // the builder's grid, stage after stage
    std::vector<uint32_t> nz_est, nz_worst; // per list: estimated / largest possible number of entries
    uint32_t nz_nrec = 0;                 // per-(list, chunk) records
    uint64_t nz_cap = 0;                  // pool capacity: the sum of the regions
    // mozjpeg's scan search (the default profile; CSH_PROFILE=plain keeps the stock script): the candidate scans are coded in stages
    // -- work items, slots, token chunks and tables of one stage behind those of the stage before -- and the host replays
    // jcmaster.c select_scans on their sizes in between.  mozjpeg codes its candidates one after the other and skips ahead as soon as
    // a decision is made; the stages follow that order: what every image needs (ST_1, ST_2), and what only an image whose search runs
    // on needs (ST_1B: luma at Al 3; ST_2B / ST_2C: the fourth and fifth frequency split) -- those stages run only when some image
    // asks for them, and then only over the work items of those images (EncCtx::work_active).
    bool search = false;
    enum { ST_1 = 0, ST_1B = 1, ST_2 = 2, ST_2B = 3, ST_2C = 4, ST_N = 5 };
    struct Stage { uint32_t work0 = 0, nwork = 0, slot0 = 0, nslots = 0, ech0 = 0, nech = 0, table0 = 0, ntables = 0, plan0 = 0, nplans = 0, nzc0 = 0, nnzc = 0, ls0 = 0, nls = 0, ts0 = 0, nts = 0; } stage[ST_N];
    void stage_begin(Stage &sg) {
        sg.work0 = uint32_t(swork.size()); sg.slot0 = nslots; sg.ech0 = uint32_t(echunks.size()); sg.table0 = uint32_t(ntables); sg.plan0 = uint32_t(plans.size());
        sg.nzc0 = uint32_t(nzchunks.size()); sg.ls0 = nlist_slots; sg.ts0 = ntok_slots;
    }
    void stage_end(Stage &sg) {
        sg.nwork = uint32_t(swork.size()) - sg.work0; sg.nslots = nslots - sg.slot0; sg.nech = uint32_t(echunks.size()) - sg.ech0;
        sg.ntables = uint32_t(ntables) - sg.table0; sg.nplans = uint32_t(plans.size()) - sg.plan0;
        sg.nnzc = uint32_t(nzchunks.size()) - sg.nzc0; sg.nls = nlist_slots - sg.ls0; sg.nts = ntok_slots - sg.ts0;
    }
    struct SearchImg {
        int cand_work[64]; int ncand;       // candidate number -> work item (-1: not coded by itself -- see search_work)
        int Al_luma = 0, Al_chroma = 0;
        uint64_t best_luma = 0, best_chroma = 0;   // running minimum of the decision in progress
        int split_luma = 0, split_chroma = 0;
        bool luma_on = false, chroma_on = false;    // the decision in progress needs the next stage's candidates
    };
    std::vector<SearchImg> simg;
    std::vector<uint8_t> work_active;               // per work item: coded in the (gated) stage about to run
    uint32_t n_gated_runs = 0;                      // how many of the conditional stages the last run needed (csh_timing.n_search_extra)
    std::vector<uint32_t> img_list, img_nlist, h_cost;
    std::map<std::array<int, 5>, int> cand_script;   // (component, Ss, Se, Ah, Al) -> EncScan index
    // mozjpeg's quantiser half (CSH_PROFILE=mozjpeg): overshoot deringing in front of every forward DCT; trellis quantisation behind it --
    // a third stage of work items (one statistics scan per component, coded for its histogram only) and the two k_trellis kernels
    bool trellis = false, dering = false;
    Stage tstage;
    std::vector<TrellisWork> twork;
    std::vector<TrellisChunk> tchunks;
    uint32_t t_units = 0, t_max_rows = 0;
    std::vector<uint32_t> trows;          // k_trellis_dc: (work item << 16 | iMCU row), longest rows first
    bool t_sort = false;                  // k_trellis_ac takes its blocks in order of list length (progressive output: the statistics lists count them)
    bool nz_once = false;                 // progressive output under the trellis quantiser: its levels go into the statistics scan's level-0 lists and the coding stages filter those (no second k_nzlist over the tiles)
    PinnedBytes bits_pool;
    std::vector<uint8_t> hdr_pool;
    std::vector<uint32_t> hdr_off;
    uint32_t ntiles = 0, ntiles_in = 0, ntiles_out = 0, max_tiles = 0, max_units = 0, max_dummy = 0;
    uint64_t total_units = 0, total_words = 0, plane_bytes = 0, oplane_bytes = 0;
    uint32_t max_quads = 0;
    int ntables = 0;
    uint64_t raw_bytes_cap = 0, out_cap = 0;

    // device buffers
    DevBuf<uint8_t> d_bits, d_clean, d_planes, d_oplanes, d_hdr, d_out, d_tail;
    DevBuf<ParScan> d_pscans;
    DevBuf<uint64_t> d_pstate, d_relax_list[2], d_unstuff_off, d_blk_off, d_dc_off;
    DevBuf<uint32_t> d_unstuff_cnt, d_nblk, d_need_seq, d_need_seq_init, d_relax_cnt, d_scan_pending, d_cut_block, d_claim;
    DevBuf<uint16_t> d_hyp;
    DevBuf<int32_t> d_dcdiff;
    DevBuf<ImgDesc> d_imgs;
    DevBuf<DecScan> d_dscans;
    DevBuf<ProgChain> d_chains;
    DevBuf<int> d_chain_scans;
    DevBuf<uint64_t> d_refine_hist;
    DevBuf<uint32_t> d_refine_pos, d_refine_prog;
    DevBuf<RefineUnit> d_refine_units;
    std::vector<RefineUnit> refine_units;
    DevBuf<DevHuffSet> d_hsets;
    DevBuf<ParHuffSet> d_phsets;
    DevBuf<ParHuffSet4> d_phsets4;
    DevBuf<DevQuant> d_quants;
    DevBuf<PlaneWork> d_pwork;
    DevBuf<ResizeWork> d_rwork;
    DevBuf<ResizeTap> d_rtaps;
    DevBuf<float> d_rweights, d_rtmp;
    DevBuf<uint8_t> d_rgb;
    DevBuf<EncScan> d_script;
    DevBuf<ScanWork> d_swork;
    DevBuf<uint32_t> d_slot_work;
    DevBuf<EChunk> d_echunks;
    DevBuf<SlotRec> d_slots;
    DevBuf<TokPlan> d_plans;
    DevBuf<int16_t> d_coef, d_dct_raw;
    DevBuf<csw::WebpImg> d_wimgs;
    DevBuf<uint8_t> d_wwork, d_wscratch;
    DevBuf<uint32_t> d_wpart, d_wstats;
    DevBuf<uint8_t> d_wprobs, d_wupdate;
    uint32_t wmax_mbh = 0;
    DevBuf<int16_t> d_wlevels;
    DevBuf<uint64_t> d_corr, d_symbits, d_eobbits, d_tok_off, d_chunk_off, d_scan_raw_off, d_img_off;
    DevBuf<uint32_t> d_tok_cursor;
    DevBuf<TokRegion> d_regions;
    DevBuf<uint16_t> d_eobrun, d_slot_hist;
    DevBuf<uint32_t> d_img_list, d_img_nlist, d_scan_cost, d_slot_raw, d_slot_eobh, d_long_runs, d_long_cnt, d_tokens, d_chunk_ntok, d_chunk_bits, d_raw, d_scan_pad, d_chunk_ff, d_hdr_off, d_img_size, d_img_size_pad, d_status, d_overflow;
    DevBuf<DevEncTable> d_tables;
    DevBuf<uint8_t> d_scan_tmp;
    DevBuf<uint8_t> d_work_active;
    DevBuf<NzList> d_nzlists;
    DevBuf<NzSet> d_nzsets;
    DevBuf<NzChunk> d_nzchunks;
    DevBuf<uint32_t> d_nz_pool, d_nz_cursor, d_nz_chunk_off, d_nz_chunk_cnt, d_list_slots, d_tok_slots;
    DevBuf<TrellisWork> d_twork;
    DevBuf<TrellisChunk> d_tchunks;
    DevBuf<uint32_t> d_trows, d_tperm;
    DevBuf<uint8_t> d_tblk_cnt;
    DevBuf<uint16_t> d_tblk_off;
    DevBuf<uint64_t> d_tlambda;
    DevBuf<uint64_t> d_tdcbt;
    DevBuf<uint32_t> d_tspill;

    std::vector<uint32_t> h_img_size;
    std::vector<uint64_t> h_img_off;
    std::vector<uint32_t> h_status;
    bool ran = false;

    // (wait for whatever is still queued -- a run that failed half-way leaves launches behind -- before the members hand their device blocks back to the cache)
    ~csh_batch() { if (have_stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); } }
};

// ------------------------------------------------------------------------------------------------
static void build_dev_huff(const HuffSpec &h, DevHuff &d) {
    memset(&d, 0, sizeof d);
    for (int l = 0; l < 18; l++) d.maxcode[l] = -1;
    if (!h.present) return;
    int code = 0, p = 0;
    for (int l = 1; l <= 16; l++) {
        if (h.bits[l]) {
            d.valptr[l] = p - code;
            for (int i = 0; i < h.bits[l]; i++, p++, code++)
                if (l <= 9) {
                    int base = code << (9 - l);
                    for (int k = 0; k < (1 << (9 - l)); k++) d.look[(base + k) & 511] = uint16_t((l << 8) | h.vals[p]);
                }
            d.maxcode[l] = code - 1;
        }
        code <<= 1;
    }
    d.maxcode[17] = 0x7FFFFFFF;
    memcpy(d.vals, h.vals, 256);
}
// two-level table of the parallel decoder; returns false when the shared sub-table pool is exhausted
static bool build_par_huff(const HuffSpec &h, uint16_t root[512], uint16_t *sub, int &sub_used) {
    memset(root, 0, 512 * sizeof(uint16_t));
    if (!h.present) return true;
    // canonical codes, left-aligned to 16 bits
    struct Code { uint16_t first; uint8_t len, sym; };
    std::vector<Code> longc;
    int code = 0, p = 0;
    for (int l = 1; l <= 16; l++) {
        for (int i = 0; i < h.bits[l]; i++, p++, code++) {
            if (l <= 9) {
                int base = code << (9 - l);
                for (int k = 0; k < (1 << (9 - l)); k++) root[(base + k) & 511] = uint16_t((l << 8) | h.vals[p]);
            } else longc.push_back({uint16_t(code << (16 - l)), uint8_t(l), h.vals[p]});
        }
        code <<= 1;
    }
    for (size_t i = 0; i < longc.size();) {
        const int prefix = longc[i].first >> 7;
        size_t j = i;
        int maxlen = 0;
        while (j < longc.size() && (longc[j].first >> 7) == prefix) { maxlen = std::max<int>(maxlen, longc[j].len); j++; }
        const int nbits = maxlen - 9, n = 1 << nbits;
        if (sub_used + n > CSH_PAR_SUB) return false;
        uint16_t *t = sub + sub_used;
        memset(t, 0, n * sizeof(uint16_t));
        for (size_t c = i; c < j; c++) {
            int lo = (longc[c].first & 127) >> (7 - nbits), span = 1 << (maxlen - longc[c].len);
            for (int k = 0; k < span; k++) t[lo + k] = uint16_t((longc[c].len << 8) | longc[c].sym);
        }
        root[prefix & 511] = uint16_t(0x8000 | (nbits << 12) | sub_used);
        sub_used += n;
        i = j;
    }
    return true;
}

static void make_quant(const uint16_t nat[64], DevQuant &q) {
    for (int k = 0; k < 64; k++) {
        q.q[k] = nat[kZigZag[k]];
        q.div[k] = int32_t(q.q[k]) * 8;
        q.rcp[k] = float((1.0 / double(q.div[k])) * (1.0 + 1.0 / 524288.0));   // the pixel kernels' quantiser: one fma (k_pixel.hip quant_one); exact for every 16-bit q
        q.lt[k] = float(1.0 / double(int(q.q[k]) * int(q.q[k])));   // mozjpeg quantize_trellis, mode 1: lambda_table[i] = 1.0 / (q * q)
        q.mul[k] = 0; q.sh[k] = 0;
        if (q.div[k] > 0 && q.div[k] < (1 << 14)) {
            int lg = 0;
            while ((2 << lg) <= q.div[k]) lg++;   // floor(log2 div)
            const int P = std::max(25, lg + 18);
            q.mul[k] = uint32_t((1ull << P) / uint64_t(q.div[k])) + 1u;
            q.sh[k] = uint32_t(32 - P);
        }
    }
}

static void fill_geom(const JComp &c, CompGeom &g, uint32_t &ntiles) {
    g.h = c.h; g.v = c.v; g.comp_w = c.comp_w; g.comp_h = c.comp_h;
    g.real_bw = c.real_bw; g.real_bh = c.real_bh; g.bw = c.bw; g.bh = c.bh;
    g.ntiles = (c.bw * c.bh + 63) / 64;
    g.tile_base = ntiles;
    ntiles += g.ntiles;
}

// output script -> EncScan entries (progressive: libjpeg jpeg_simple_progression; sequential: one interleaved scan).
// Y uses Huffman table ids 0, chroma ids 1.
static void add_script(std::vector<EncScan> &v, int ncomp, bool progressive) {
    for (const OutScan &o : output_script(ncomp, progressive)) {
        EncScan e;
        memset(&e, 0, sizeof e);
        e.ncomp = o.ncomp; e.Ss = o.Ss; e.Se = o.Se; e.Ah = o.Ah; e.Al = o.Al;
        for (int k = 0; k < o.ncomp; k++) e.comp[k] = o.comp[k];
        if (!progressive) {
            e.sequential = 1;
            for (int k = 0; k < o.ncomp; k++) {   // DHT order: per component DC then AC, each table once (libjpeg write_scan_header)
                int id = o.comp[k] ? 1 : 0;
                int di = -1, ai = -1;
                for (int t = 0; t < e.ntables; t++) { if (e.dht_id[t] == id) di = t; if (e.dht_id[t] == (0x10 | id)) ai = t; }
                if (di < 0) { di = e.ntables; e.dht_id[e.ntables++] = id; }
                if (ai < 0) { ai = e.ntables; e.dht_id[e.ntables++] = 0x10 | id; }
                e.dc_tbl[k] = di; e.ac_tbl[k] = ai;
                e.sos_tdta[k] = (id << 4) | id;
            }
        } else if (o.Ss == 0) {
            if (o.Ah == 0) {
                e.ntables = 0;
                for (int k = 0; k < o.ncomp; k++) {
                    int id = o.comp[k] ? 1 : 0;
                    int idx = -1;
                    for (int t = 0; t < e.ntables; t++) if (e.dht_id[t] == id) idx = t;
                    if (idx < 0) { idx = e.ntables; e.dht_id[e.ntables++] = id; }
                    e.dc_tbl[k] = idx;
                    e.sos_tdta[k] = id << 4;
                }
            }
        } else {
            int id = o.comp[0] ? 1 : 0;
            e.ntables = 1; e.dht_id[0] = 0x10 | id; e.sos_tdta[0] = id;
        }
        v.push_back(e);
    }
}

static bool is_jpeg(const uint8_t *d, size_t n) { return n >= 3 && d[0] == 0xFF && d[1] == 0xD8 && d[2] == 0xFF; }
static int sniff_type(const uint8_t *d, size_t n) {
    if (is_jpeg(d, n)) return CS_TYPE_JPEG;
    if (n >= 8 && !memcmp(d, "\x89PNG\r\n\x1a\n", 8)) return CS_TYPE_PNG;
    if (n >= 12 && !memcmp(d, "RIFF", 4) && !memcmp(d + 8, "WEBP", 4)) return CS_TYPE_WEBP;
    if (n >= 6 && (!memcmp(d, "GIF87a", 6) || !memcmp(d, "GIF89a", 6))) return CS_TYPE_GIF;
    if (n >= 4 && (!memcmp(d, "II*\0", 4) || !memcmp(d, "MM\0*", 4))) return CS_TYPE_TIFF;
    return CS_TYPE_UNKN;
}

// decide the output frame for one parsed JPEG; returns 0 or an error code
// image-rs Lanczos3 taps of one axis (imageops::sample) -- host side, same libm calls as the oracle
static float sincf_(float t) { float a = t * 3.14159265358979323846f; return t == 0.0f ? 1.0f : sinf(a) / a; }
static float lanczos3f(float x) { return fabsf(x) < 3.0f ? sincf_(x) * sincf_(x / 3.0f) : 0.0f; }
void csh_lanczos_axis(int in_size, int out_size, bool identity, std::vector<ResizeTap> &taps, std::vector<float> &weights) {   // also used by png_pipeline.cpp (resize_host.h)
    for (int o = 0; o < out_size; o++) {
        ResizeTap t;
        t.woff = uint32_t(weights.size());
        if (identity) { t.left = o; t.n = 1; weights.push_back(1.0f); taps.push_back(t); continue; }
        float ratio = float(in_size) / float(out_size);
        float sratio = ratio < 1.0f ? 1.0f : ratio;
        float support = 3.0f * sratio;
        float center = (float(o) + 0.5f) * ratio;
        long left = long(floorf(center - support)); if (left < 0) left = 0; if (left > in_size - 1) left = in_size - 1;
        long right = long(ceilf(center + support)); if (right < left + 1) right = left + 1; if (right > in_size) right = in_size;
        center = center - 0.5f;
        float sum = 0.0f;
        for (long i = left; i < right; i++) { float w = lanczos3f((float(i) - center) / sratio); weights.push_back(w); sum += w; }
        for (size_t i = t.woff; i < weights.size(); i++) weights[i] /= sum;
        t.left = int(left); t.n = int(right - left);
        taps.push_back(t);
    }
}
// libcaesium resize.rs compute_dimensions [UPSTREAM-RECALL]: both given -> exact; one given -> keep aspect, f32, round half away
void csh_compute_dimensions(int ow, int oh, int dw, int dh, int &nw, int &nh) {
    if (dw > 0 && dh > 0) { nw = dw; nh = dh; }
    else {
        float ratio = float(ow) / float(oh);
        if (dw > 0) { nw = dw; nh = int(roundf(float(dw) / ratio)); }
        else { nh = dh; nw = int(roundf(float(dh) * ratio)); }
    }
    if (nw < 1) nw = 1;
    if (nh < 1) nh = 1;
}

static int plan_item(Item &it, const CCSParameters &p, bool lossless) {
    const JpegInfo &in = it.in;
    if (in.ncomp != 1 && in.ncomp != 3) { it.msg = "unsupported component count (CMYK/YCCK not on the device path yet)"; return CS_ERR_JPEG_FEATURE; }
    if (in.ncomp == 3) {
        bool rgb_ids = in.comp[0].id == 'R' && in.comp[1].id == 'G' && in.comp[2].id == 'B';
        if (in.adobe_transform == 0 || rgb_ids) { it.msg = "RGB-colourspace JPEG not on the device path yet"; return CS_ERR_JPEG_FEATURE; }
    }
    // one image's planes, tiles and unit arrays are indexed with 32 bits and take ~25 bytes per pixel of HBM: a header that declares
    // more than 2^28 pixels (16384 x 16384) fails here, by itself, instead of sizing the whole batch's pools
    if (uint64_t(in.width) * uint64_t(in.height) > (1ull << 28)) { it.msg = "image dimensions too large for the device path"; return CS_ERR_JPEG_FEATURE; }
    it.out = JpegInfo();
    JpegInfo &o = it.out;
    o.width = in.width; o.height = in.height; o.ncomp = in.ncomp;
    if (p.width || p.height) {
        if (lossless) { it.msg = "resize + lossless transcode not on the device path"; return CS_ERR_UNSUPPORTED; }
        csh_compute_dimensions(in.width, in.height, int(p.width), int(p.height), o.width, o.height);
        if (o.width > 65500 || o.height > 65500 || uint64_t(o.width) * uint64_t(o.height) > (1ull << 28)) { it.msg = "resize target too large for JPEG"; return CS_ERR_JPEG_FEATURE; }
    }
    if (lossless) {
        for (int c = 0; c < in.ncomp; c++) o.comp[c] = in.comp[c];
        jpeg_geometry(o);
        return 0;
    }
    int ss = int(p.jpeg_chroma_subsampling);
    if (ss == 0) ss = 420;
    for (int c = 0; c < in.ncomp; c++) { o.comp[c].id = c + 1; o.comp[c].h = o.comp[c].v = 1; o.comp[c].tq = c ? 1 : 0; }
    if (in.ncomp == 3) {
        bool in444 = in.comp[0].h == 1 && in.comp[0].v == 1, in420 = in.comp[0].h == 2 && in.comp[0].v == 2;
        bool in422 = in.comp[0].h == 2 && in.comp[0].v == 1;
        bool chroma11 = in.comp[1].h == 1 && in.comp[1].v == 1 && in.comp[2].h == 1 && in.comp[2].v == 1;
        if (!chroma11 || !(in444 || in420 || in422)) { it.msg = "input chroma sampling other than 4:4:4 / 4:2:2 / 4:2:0 not on the device path yet"; return CS_ERR_JPEG_FEATURE; }
        if (ss == 420) { o.comp[0].h = 2; o.comp[0].v = 2; }
        else if (ss == 422) { o.comp[0].h = 2; o.comp[0].v = 1; }
        else if (ss != 444) { it.msg = "output chroma subsampling 4:1:1 not on the device path yet"; return CS_ERR_JPEG_FEATURE; }
    }
    jpeg_geometry(o);
    return 0;
}

// Device pools and pinned blocks of finished batches stay in per-device / process-wide caches (devmem.hpp: hipMalloc / hipFree cost more than
// the kernels); a caller that wants the memory back -- another process is about to use the device -- says so here.
extern "C" void csh_release_cached_memory(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    for (int d = 0; d < n && d < 64; d++) device_cache(d).trim(0);
    pinned_cache().trim(0);
}
// A process that knows it will use a device says so early, from a thread of its own: runtime start-up, the device's context and the library's code
// objects (the first launch loads them) take ~0.15 s that can pass while the caller is still reading its files (the CLI does: cli.cpp)
__global__ void k_warmup(uint32_t *p) { if (p && threadIdx.x == 1024) *p = 0; }
extern "C" void csh_warmup(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n || hipSetDevice(device) != hipSuccess) return;
    hipStream_t st;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return;
    CSH_LAUNCH(k_warmup, dim3(1), dim3(1), st, static_cast<uint32_t *>(nullptr));
    (void)hipStreamSynchronize(st);
    (void)hipStreamDestroy(st);
}
extern "C" int csh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int batch_create(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, bool webp, csh_batch **out, bool rgb_out = false, const csp_pixels *px = nullptr);
extern "C" int csh_batch_create(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, csh_batch **out) { return batch_create(inputs, count, p, device, false, out); }
// JPEG in, pixels out (the front half of convert_in_memory to PNG): decode and resize only; the RGB stays in device memory (csh_batch_pixels)
extern "C" int csh_batch_create_pixels(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, csh_batch **out) { return batch_create(inputs, count, p, device, false, out, true); }
// JPEG in, WebP out (caesium::convert_in_memory to SupportedFileTypes::WebP, compressor.rs:289,300): same decode and resize, then the VP8 encoder
extern "C" int csh_batch_create_webp(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, csh_batch **out) { return batch_create(inputs, count, p, device, true, out); }
// token pool (k_entropy.hip): every region gets its estimate x tok_scale (the scale grows on overflow, like the other pools)
static void layout_token_pool(csh_batch *b) {
    b->regions.resize(b->region_est.size());
    // CSH_TEST_POOL_SHIFT=n (tests): every estimate divided by 2^n, so that the first runs overflow and the batch goes through its retries with larger pools
    // (read once per batch object, at its first layout: the retries of a run keep what the run started with -- ADVICE r05; an unsupported test hook)
    if (b->test_pool_shift < 0) b->test_pool_shift = getenv("CSH_TEST_POOL_SHIFT") ? std::min(16, std::max(0, atoi(getenv("CSH_TEST_POOL_SHIFT")))) : 0;
    const int shift = b->test_pool_shift;
    uint64_t at = 0;
    for (size_t i = 0; i < b->regions.size(); i++) {
        const uint64_t cap = std::min<uint64_t>(std::max<uint64_t>((uint64_t(b->region_est[i]) * b->tok_scale) >> shift, 64), 0xFFFFFFF0ull);
        b->regions[i].base = at; b->regions[i].cap = uint32_t(cap); b->regions[i].pad = 0;
        at += (cap + 3) & ~uint64_t(3);
    }
    b->tok_cap = at + 64;
    // the list pool (k_aclist.hip): the same rule, capped by what a list can hold at most; regions start on 16-byte boundaries
    at = 0;
    for (size_t i = 0; i < b->nzlists.size(); i++) {
        const uint64_t cap = std::min<uint64_t>(std::max<uint64_t>((uint64_t(b->nz_est[i]) * b->tok_scale) >> shift, 64), b->nz_worst[i]);
        b->nzlists[i].base = at; b->nzlists[i].cap = uint32_t(cap);
        at += (cap + 3) & ~uint64_t(3);
    }
    b->nz_cap = at + 64;
}
static int batch_create(const CByteArray *inputs, size_t count, const CCSParameters *p, int device, bool webp, csh_batch **out, bool rgb_out, const csp_pixels *px) {
    *out = nullptr;
    if (csh_device_count() <= device) { csh_set_error("no HIP device %d available (libcaesium_hip has no CPU path)", device); return CS_ERR_NO_DEVICE; }
    if (hipSetDevice(device) != hipSuccess) { csh_set_error("hipSetDevice(%d) failed", device); return CS_ERR_NO_DEVICE; }
    if (count > 6000) { csh_set_error("csh_batch_create: at most 6000 files per device batch (cs_batch_compress splits for you)"); return CS_ERR_POOL_OVERFLOW; }
    // CSH_TRACE: host-side laps of this call on stderr (what the boundary pays in front of the first kernel)
    const bool trace = getenv("CSH_TRACE") != nullptr;
    auto lap_t = std::chrono::steady_clock::now();
    std::string laps;
    auto lap = [&](const char *what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        char buf[64];
        snprintf(buf, sizeof buf, " %s %.1f", what, std::chrono::duration<double, std::milli>(now - lap_t).count());
        laps += buf; lap_t = now;
    };
    std::unique_ptr<csh_batch> b(new csh_batch);
    b->device = device;
    b->params = *p;
    b->lossless = p->jpeg_optimize && !webp && !rgb_out && !px;
    b->webp = webp; b->rgb_out = rgb_out;
    const bool progressive = p->jpeg_progressive;
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) { csh_set_error("hipStreamCreate failed"); return CS_ERR_NO_DEVICE; }
    b->have_stream = true;
    b->items.resize(count);

    uint16_t qout_nat[64];
    quality_table(int(p->jpeg_quality), qout_nat);
    DevQuant qo; make_quant(qout_nat, qo);
    b->quants.push_back(qo);  // index 0: output table (luma == chroma in mozjpeg's profile 3)
    b->q_base = int(b->quants.size()) - 1;   // then one table per quality 1..100 (size targeting re-targets per image)
    for (int q = 1; q <= 100; q++) { uint16_t tq[64]; quality_table(q, tq); DevQuant dq; make_quant(tq, dq); b->quants.push_back(dq); }
    b->progressive = progressive;

    add_script(b->script, 3, true);   // entries 0..9
    add_script(b->script, 1, true);   // entries 10..15
    add_script(b->script, 3, false);  // entry 16
    add_script(b->script, 1, false);  // entry 17
    const int script_base3 = progressive ? 0 : 16, script_base1 = progressive ? 10 : 17;
    // CSH_PROFILE names what stands in for libcaesium's JPEG engine.  Unset or "mozjpeg": the whole JCP_MAX_COMPRESSION profile libcaesium's
    // -q runs (compressor.rs:415,427): scan search + trellis quantisation + overshoot deringing.  "scalar": the scan search over the scalar
    // quantiser (the pieces pinned by samples/j0.JPG and libjpeg-turbo); "plain": the stock jpeg_simple_progression script over the
    // scalar quantiser (whole files equal libjpeg-turbo's); "mozjpeg-trellis" / "mozjpeg-dering": one half of the quantiser each (scan search kept)
    const char *penv = getenv("CSH_PROFILE");
    const std::string profile = penv && *penv ? penv : "mozjpeg";
    b->search = progressive && !webp && !rgb_out && profile != "plain";
    const bool lossy_jpeg = !b->lossless && !webp && !rgb_out;
    b->trellis = lossy_jpeg && (profile == "mozjpeg" || profile == "mozjpeg-trellis");
    b->dering = lossy_jpeg && (profile == "mozjpeg" || profile == "mozjpeg-dering");
    {   // (CSH_NZ_ONCE=0: the coding stages build their level-0 lists from the coefficient tiles again, as before round 5; CSH_TR_SORT=0 implies it -- the per-block list offsets come with the sort's counts)
        const char *once = getenv("CSH_NZ_ONCE"), *ts = getenv("CSH_TR_SORT");
        b->nz_once = b->trellis && progressive && !(once && !strcmp(once, "0")) && !(ts && !strcmp(ts, "0"));
    }
    // EncScan entries of the search's candidates, made on first use
    auto cand_index = [&](int comp, int Ss, int Se, int Ah, int Al) -> int {
        const std::array<int, 5> key = {comp, Ss, Se, Ah, Al};
        auto f = b->cand_script.find(key);
        if (f != b->cand_script.end()) return f->second;
        EncScan e;
        memset(&e, 0, sizeof e);
        const int id = comp ? 1 : 0;
        e.ncomp = 1; e.comp[0] = comp; e.Ss = Ss; e.Se = Se; e.Ah = Ah; e.Al = Al;
        e.ntables = 1; e.dht_id[0] = 0x10 | id; e.sos_tdta[0] = id;
        b->script.push_back(e);
        return b->cand_script[key] = int(b->script.size()) - 1;
    };
    auto dc_scan_index = [&](int ncomp) -> int {   // first DC scan of all components at Al = 0 (dc_scan_opt_mode 0)
        const std::array<int, 5> key = {-ncomp, 0, 0, 0, 0};
        auto f = b->cand_script.find(key);
        if (f != b->cand_script.end()) return f->second;
        EncScan e;
        memset(&e, 0, sizeof e);
        e.ncomp = ncomp;
        for (int k = 0; k < ncomp; k++) {
            e.comp[k] = k;
            const int id = k ? 1 : 0;
            int idx = -1;
            for (int t = 0; t < e.ntables; t++) if (e.dht_id[t] == id) idx = t;
            if (idx < 0) { idx = e.ntables; e.dht_id[e.ntables++] = id; }
            e.dc_tbl[k] = idx; e.sos_tdta[k] = id << 4;
        }
        b->script.push_back(e);
        return b->cand_script[key] = int(b->script.size()) - 1;
    };
    // the NzList of (image, component, Al), made on first use (k_aclist.hip): the list itself, level 0 (what the others are filtered from), the set
    auto nz_list = [&](int img_index, int comp, int Al, const ImgDesc &im, size_t in_len) -> uint32_t {
        if (b->nzset_of.size() < size_t(img_index + 1) * CSH_MAX_COMPS) b->nzset_of.resize(size_t(img_index + 1) * CSH_MAX_COMPS, -1);
        int &si = b->nzset_of[size_t(img_index) * CSH_MAX_COMPS + size_t(comp)];
        const uint32_t nu = uint32_t(im.out[comp].real_bw * im.out[comp].real_bh);
        if (si < 0) {
            NzSet S;
            memset(&S, 0, sizeof S);
            for (int L = 0; L < CSH_NZ_LEVELS; L++) S.list[L] = 0xFFFFFFFFu;
            S.cnt_base = 0xFFFFFFFFu;
            S.nunits = nu; S.real_bw = im.out[comp].real_bw; S.bw = im.out[comp].bw; S.tile_base = im.out[comp].tile_base;   // rebased with the plans below
            si = int(b->nzsets.size());
            b->nzsets.push_back(S); b->nzset_built.push_back(0u); b->nzset_comp.push_back(comp); b->nzset_image.push_back(img_index);
        }
        uint64_t blocks_all = 0;
        for (int k = 0; k < im.ncomp; k++) blocks_all += uint64_t(im.out[k].real_bw) * im.out[k].real_bh;
        // a non-zero coefficient costs a source file 3 bits at the very least and ~5 at ordinary qualities; one of magnitude >= 2^Al more
        static const uint32_t kShare[CSH_NZ_LEVELS] = {20, 16, 12, 9};   // eighths of the file's bytes, per level
        for (int L : {0, Al}) {
            if (b->nzsets[size_t(si)].list[L] != 0xFFFFFFFFu) continue;
            const uint32_t nch = (nu + 255) / 256;
            NzList R;
            memset(&R, 0, sizeof R);
            R.chunk0 = b->nz_nrec; b->nz_nrec += nch;
            b->nzsets[size_t(si)].list[L] = uint32_t(b->nzlists.size());
            b->nzlists.push_back(R);
            const uint64_t est = uint64_t(in_len) * kShare[L] / 8 * nu / std::max<uint64_t>(1, blocks_all) + nu + 4ull * nch + 256;
            const uint64_t worst = uint64_t(nu) * 64 + 4ull * nch;
            b->nz_est.push_back(uint32_t(std::min<uint64_t>(est, worst)));
            b->nz_worst.push_back(uint32_t(std::min<uint64_t>(worst, 0xFFFFFFF0ull)));
        }
        return b->nzsets[size_t(si)].list[Al];
    };
    // work items, slots, token chunks and plans of one image for a list of scans (EncScan indices), in list order
    auto add_works = [&](Item &it, ImgDesc &im, int img_index, const std::vector<int> &list, size_t in_len, const JpegInfo &o, bool stats_only = false) {
        const int w_first = int(b->swork.size());
        uint32_t nz_need[CSH_MAX_COMPS] = {0, 0, 0}, nz_gate[CSH_MAX_COMPS] = {0, 0, 0};
        for (int sidx : list) {
            const EncScan &e = b->script[sidx];
            ScanWork w;
            memset(&w, 0, sizeof w);
            w.image = img_index; w.scan = sidx;
            w.out_off = 0xFFFFFFFFu;   // not part of a file until k_layout says so (a conditional stage of the scan search may never run)
            // progressive AC first-pass scans are coded from the component's compacted list at their Al; everything else from tokens
            const bool from_list = e.Ss > 0 && !e.sequential && e.Ah == 0;
            w.list = 0xFFFFFFFFu;
            if (from_list) {
                if (e.Al >= CSH_NZ_LEVELS) { it.code = CS_ERR_JPEG_FEATURE; it.msg = "internal: output scan script outside what the list coder carries"; }
                else {
                    w.list = nz_list(img_index, e.comp[0], e.Al, im, in_len);
                    if (!nz_need[e.comp[0]]) nz_gate[e.comp[0]] = uint32_t(b->swork.size());
                    nz_need[e.comp[0]] |= 1u << e.Al;
                }
            }
            if (e.Ss == 0 && e.ncomp > 1) w.nunits = uint32_t(im.omcus_x * im.omcus_y);
            else w.nunits = uint32_t(im.out[e.comp[0]].real_bw * im.out[e.comp[0]].real_bh);
            w.unit_base = uint32_t(b->total_units);
            b->total_units += w.nunits;
            w.corr_base = 0xFFFFFFFFu;
            if (e.Ss > 0 && !e.sequential && e.Ah) { w.corr_base = uint32_t(b->total_corr); b->total_corr += w.nunits; }
            w.word_base = uint32_t(b->total_words);
            if (e.Ss) b->total_words += (w.nunits + 63) / 64;
            w.table_base = uint32_t(b->ntables);
            b->ntables += e.ntables;
            b->max_units = std::max(b->max_units, w.nunits);
            {   // its slots, one per 256 units: counted here, written by k_make_slots (k_aclist.hip) from this record
                const uint32_t nch = (w.nunits + 255) / 256;
                w.first_chunk = b->nslots; b->nslots += nch;
                w.hist_row0 = b->hist_rows; b->hist_rows += nch * uint32_t(e.ntables);
                uint32_t &cursor = w.list != 0xFFFFFFFFu ? b->nlist_slots : b->ntok_slots;
                w.ls_base = cursor; cursor += nch;
            }
            if (e.Ss == 0 || e.sequential) {   // DC scans and sequential-mode scans: one token workgroup per (scan, 256 units)
                // tokens of a DC scan are known exactly (one per block, or one per fifteen blocks' bits); a sequential-mode block has at most 64 + 3
                uint32_t blocks = 0;
                for (int k = 0; k < e.ncomp; k++) blocks += e.ncomp > 1 ? uint32_t(o.comp[e.comp[k]].h * o.comp[e.comp[k]].v) : 1u;
                const uint32_t per_unit = e.sequential ? blocks * 20u : (e.Ah ? (blocks + 14u) / 15u : blocks);
                for (uint32_t j = 0; j < (w.nunits + 255) / 256; j++) b->echunks.push_back(EChunk{uint32_t(b->swork.size()), 0, 1, j, 0, uint32_t(b->region_est.size())});
                b->region_est.push_back(stats_only ? 64u : w.nunits * per_unit + 64);
            }
            b->swork.push_back(w);
        }
        // the lists this call's scans need and no earlier stage makes: one builder wave per 256 blocks (the trellis stage's statistics
        // scans are coded from the scalar-quantised coefficients, the other stages from the trellis's: that stage makes its level 0 anew)
        for (int c = 0; c < im.ncomp; c++) {
            if (!nz_need[c]) continue;
            const int si = b->nzset_of[size_t(img_index) * CSH_MAX_COMPS + size_t(c)];
            uint32_t levels = stats_only ? (nz_need[c] | 1u) : ((nz_need[c] | 1u) & ~b->nzset_built[size_t(si)]);
            b->nzset_built[size_t(si)] |= levels;
            // the trellis stage's statistics list takes the trellis's levels (k_trellis_ac): a coding stage does not build level 0 from the tiles, it drops the list's zero entries
            if (!stats_only && b->nz_once && (levels & 1u)) levels = (levels & ~1u) | CSH_NZ_COMPACT0;
            const uint32_t nu = b->nzsets[size_t(si)].nunits;
            for (uint32_t j = 0; levels && j < (nu + 255) / 256; j++) b->nzchunks.push_back(NzChunk{uint32_t(si), j, levels, nz_gate[c]});
        }
        // the progressive AC refinement scans of a component share one pass over its blocks (k_tokens)
        for (int c = 0; c < im.ncomp; c++) {
            int nac = 0;
            for (int sidx : list) { const EncScan &e = b->script[sidx]; if (e.Ss > 0 && !e.sequential && e.Ah && e.comp[0] == c) { nac++; if (e.Al > 3) nac = 99; } }
            if (nac > CSH_TK_MAXSLOT) { it.code = CS_ERR_JPEG_FEATURE; it.msg = "internal: output scan script outside what the token kernel carries"; }
            const uint32_t nu = uint32_t(im.out[c].real_bw * im.out[c].real_bh);
            if (!nac || nac > CSH_TK_MAXSLOT) continue;
            TokPlan P;
            memset(&P, 0, sizeof P);
            P.nunits = nu; P.real_bw = im.out[c].real_bw; P.bw = im.out[c].bw; P.tile_base = im.out[c].tile_base;   // tile_base of re-quantised tiles is rebased below
            P.work0 = 0xFFFFFFFFu;
            for (size_t k = 0; k < list.size(); k++) {
                const EncScan &e = b->script[list[k]];
                if (!(e.Ss > 0 && !e.sequential && e.Ah && e.comp[0] == c)) continue;
                const ScanWork &w = b->swork[size_t(w_first) + k];
                if (P.work0 == 0xFFFFFFFFu) P.work0 = uint32_t(w_first) + uint32_t(k);   // the plan's scans are coded or skipped together: the first stands for all
                AcSlot &a = P.s[P.nslot++];
                a.unit_base = w.unit_base; a.word_base = w.word_base; a.first_chunk = w.first_chunk; a.table_base = w.table_base; a.nunits_work = w.nunits; a.corr_base = w.corr_base;
                a.Ss = uint8_t(e.Ss); a.Se = uint8_t(e.Se); a.Ah = uint8_t(e.Ah); a.Al = uint8_t(e.Al);
            }
            // every non-zero coefficient becomes a token in exactly one scan of a script (~5 bits of a source file each), plus an EOB per
            // block and scan; the search's lists hold several scripts' worth
            uint64_t blocks_all = 0;
            for (int k = 0; k < im.ncomp; k++) blocks_all += uint64_t(im.out[k].real_bw) * im.out[k].real_bh;
            const uint64_t scripts = b->search ? uint64_t(nac + 1) / 2 : 1;
            const uint64_t est = uint64_t(in_len) * 3 * scripts * nu / std::max<uint64_t>(1, blocks_all) + uint64_t(nu) * nac + 1024;
            for (uint32_t j = 0; j < (nu + 255) / 256; j++) b->echunks.push_back(EChunk{uint32_t(img_index), uint16_t(c), 0, j, uint32_t(b->plans.size()), uint32_t(b->region_est.size())});
            b->region_est.push_back(stats_only ? 64u : uint32_t(std::min<uint64_t>(est, 0x3FFFFFFFu)));
            b->plan_comp.push_back(c); b->plan_image.push_back(img_index);
            b->plans.push_back(P);
        }
    };

    std::vector<std::pair<std::vector<uint8_t>, int>> hset_keys;
    std::map<std::vector<uint16_t>, int> quant_index;
    uint64_t plane_off = 64, oplane_off = 0;  // 64-bit: a resize batch of 1024 1080p files has 7 GB of planes; 64 bytes in front of the first plane: k_resample_fdct_420 reads a row's window from four bytes before it

    {   // marker parsing is per file and touches every byte of it once (the hunt for the end of each scan): all cores
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (size_t n; (n = next++) < count;) {
                Item &it = b->items[n];
                it.file_size = inputs[n].length;
                int type = sniff_type(inputs[n].data, inputs[n].length);
                if (type == CS_TYPE_UNKN) { it.code = CS_ERR_UNKNOWN_TYPE; it.msg = "unknown file type"; }
                else if (type != CS_TYPE_JPEG) { it.code = CS_ERR_UNSUPPORTED; it.msg = "this input format has no device path in this build (built: JPEG, PNG)"; }
                else it.code = parse_jpeg(inputs[n].data, inputs[n].length, it.in, it.msg);
            }
        };
        size_t nthreads = std::min<size_t>(std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency())), (count + 15) / 16);
        std::vector<std::thread> pool;
        for (size_t t = 1; t < nthreads; t++) pool.emplace_back(worker);
        worker();
        for (auto &t : pool) t.join();
    }
    lap("stream+parse");
    {
        size_t total = 0;
        for (size_t n = 0; n < count; n++) total += inputs[n].length;
        // DecScan / ParScan address the pool with 32 bits (types.h): a group whose entropy data (plus the per-restart-interval copies)
        // could pass 4 GiB is refused as a whole -- cs_batch_extent() sizes groups well below this, so only a direct caller sees it
        if (total + total / 16 + (64u << 10) >= 0xF0000000ull) { csh_set_error("csh_batch_create: more than 3.75 GiB of input in one device batch (cs_batch_extent sizes groups)"); return CS_ERR_POOL_OVERFLOW; }
        if (!b->bits_pool.reserve(total + total / 16 + (64u << 10))) { csh_set_error("out of pinned host memory"); return CS_ERR_NO_DEVICE; }
    }
    lap("pinned");
    for (size_t n = 0; n < count; n++) {
        Item &it = b->items[n];
        const uint8_t *d = inputs[n].data;
        if (it.code) continue;
        it.code = plan_item(it, *p, b->lossless);
        if (it.code) continue;

        ImgDesc im;
        memset(&im, 0, sizeof im);
        const JpegInfo &in = it.in, &o = it.out;
        im.width = in.width; im.height = in.height; im.ncomp = in.ncomp;
        im.mcus_x = in.mcus_x; im.mcus_y = in.mcus_y;
        im.progressive_in = in.progressive;
        for (int c = 0; c < in.ncomp; c++) fill_geom(in.comp[c], im.in[c], b->ntiles_in);
        for (int c = 0; c < in.ncomp; c++) {
            if (b->lossless) im.out[c] = im.in[c];
            else fill_geom(o.comp[c], im.out[c], b->ntiles_out);  // rebased behind all decoded tiles below
            im.comp_id[c] = o.comp[c].id;
            std::vector<uint16_t> key(in.qt[in.comp[c].tq], in.qt[in.comp[c].tq] + 64);
            auto f = quant_index.find(key);
            if (f == quant_index.end()) {
                DevQuant q; make_quant(key.data(), q);
                f = quant_index.emplace(key, int(b->quants.size())).first;
                b->quants.push_back(q);
            }
            im.qt_in[c] = f->second;
            im.qt_out[c] = 0;
            b->max_tiles = std::max<uint32_t>(b->max_tiles, std::max(im.in[c].ntiles, im.out[c].ntiles));
            uint32_t nd = uint32_t(im.out[c].bw * im.out[c].bh - im.out[c].real_bw * im.out[c].real_bh);
            b->max_dummy = std::max(b->max_dummy, nd);
        }
        im.omcus_x = o.mcus_x; im.omcus_y = o.mcus_y;
        const bool resized = ((p->width || p->height) && !b->lossless) || b->webp || b->rgb_out || px;   // the WebP encoder takes the RGB the resize branch produces (a plain copy at equal size)
        im.enc_w = o.width; im.enc_h = o.height;
        for (int c = 0; c < in.ncomp; c++) im.src[c] = im.in[c];
        int img_index = int(b->imgs.size());
        // entropy-coded scans
        im.first_scan = int(b->dscans.size());
        im.nscans_in = int(in.scans.size());
        for (const JScan &js : in.scans) {
            DecScan ds;
            memset(&ds, 0, sizeof ds);
            ds.bits_off = uint32_t(b->bits_pool.size());
            ds.bits_len = uint32_t(js.data_len);
            if (!b->bits_pool.append_aligned(d + js.data_off, js.data_len)) { csh_set_error("out of pinned host memory"); return CS_ERR_NO_DEVICE; }
            ds.ncomp = js.ncomp;
            if (js.ncomp > CSH_MAX_COMPS) { it.code = CS_ERR_JPEG_FEATURE; it.msg = "scan with more than 3 components"; break; }
            for (int k = 0; k < js.ncomp; k++) { ds.comp[k] = js.comp_idx[k]; ds.td[k] = js.td[k]; ds.ta[k] = js.ta[k]; }
            ds.Ss = js.Ss; ds.Se = js.Se; ds.Ah = js.Ah; ds.Al = js.Al;
            ds.restart_interval = in.restart_interval;
            ds.par_index = -1;
            // Huffman set, de-duplicated by content
            std::vector<uint8_t> key;
            for (int t = 0; t < 4; t++)
                for (const HuffSpec *h : {&js.dc[t], &js.ac[t]}) {
                    key.push_back(h->present);
                    if (h->present) { key.insert(key.end(), h->bits, h->bits + 17); key.insert(key.end(), h->vals, h->vals + h->nvals); }
                }
            int found = -1;
            for (auto &kv : hset_keys) if (kv.first == key) { found = kv.second; break; }
            if (found < 0) {
                found = int(b->hsets.size());
                DevHuffSet hs;
                for (int t = 0; t < 4; t++) { build_dev_huff(js.dc[t], hs.dc[t]); build_dev_huff(js.ac[t], hs.ac[t]); }
                b->hsets.push_back(hs);
                ParHuffSet phs;
                memset(&phs, 0, sizeof phs);
                int sub_used = 0;
                bool fits = true;
                for (int t = 0; t < 4; t++) fits = fits && build_par_huff(js.dc[t], phs.root[t], phs.sub, sub_used) && build_par_huff(js.ac[t], phs.root[4 + t], phs.sub, sub_used);
                b->phsets.push_back(phs);
                b->phset_fits.push_back(fits ? 1 : 0);
                ParHuffSet4 ph4;
                memset(&ph4, 0, sizeof ph4);
                std::array<int8_t, 8> slots;
                slots.fill(-1);
                int nslot = 0, sub4 = 0;
                bool fits4 = true;
                for (int t = 0; t < 8 && fits4; t++) {
                    const HuffSpec &h = t < 4 ? js.dc[t] : js.ac[t - 4];
                    if (!h.present) continue;
                    if (nslot == 4) { fits4 = false; break; }
                    uint16_t tmp_sub[CSH_PAR_SUB];
                    int used = 0;
                    if (!build_par_huff(h, ph4.root[nslot], tmp_sub, used) || sub4 + used > CSH_PAR_SUB4) { fits4 = false; break; }
                    for (int e = 0; e < 512; e++) if (ph4.root[nslot][e] & 0x8000u) ph4.root[nslot][e] = uint16_t(ph4.root[nslot][e] + sub4);   // rebase the sub-table offsets
                    memcpy(ph4.sub + sub4, tmp_sub, used * sizeof(uint16_t));
                    sub4 += used;
                    slots[t] = int8_t(nslot++);
                }
                if (!fits4) b->use4 = false;
                b->phsets4.push_back(ph4);
                b->slot4.push_back(slots);
                hset_keys.emplace_back(key, found);
            }
            ds.huff_set = found;
            // tables the scan needs must exist
            for (int k = 0; k < js.ncomp; k++) {
                bool need_dc = in.progressive ? (js.Ss == 0 && js.Ah == 0) : true;
                bool need_ac = in.progressive ? (js.Ss != 0) : true;
                if ((need_dc && !js.dc[js.td[k]].present) || (need_ac && !js.ac[js.ta[k]].present)) { it.code = CS_ERR_BAD_JPEG; it.msg = "scan refers to a missing Huffman table"; }
            }
            b->dscans.push_back(ds);
        }
        if (it.code) { b->dscans.resize(im.first_scan); continue; }

        // sequential-mode scans go to the parallel self-synchronising decoder: a scan without restart markers as one
        // ParScan, a scan with them as one ParScan per restart interval (each interval is an independent stream whose DC
        // prediction starts at zero -- exactly what a ParScan is); the intervals are copied into the pool once more, each
        // 64-byte aligned, because the unstuffing pass works on aligned segments
        bool par_ok = !in.progressive;
        for (size_t s = 0; s < in.scans.size(); s++) if (!b->phset_fits[b->dscans[im.first_scan + s].huff_set]) par_ok = false;
        for (const JScan &js : in.scans) if (js.data_len >= (1u << 28)) par_ok = false;
        if (uint64_t(in.mcus_x) * uint64_t(in.mcus_y) * 10 >= (1u << 24)) par_ok = false;   // k_decode_par.hip uses 24-bit multiplies on block counts
        {   // a component coded by two scans (malformed, but libjpeg decodes it: the later scan wins) must not be written by two
            // segments at once: leave the order to the sequential kernel
            int seen[4] = {0, 0, 0, 0};
            for (const JScan &js : in.scans) for (int k = 0; k < js.ncomp; k++) if (seen[js.comp_idx[k] & 3]++) par_ok = false;
        }
        struct Piece { size_t off, len; uint32_t first_mcu, nmcus; };
        std::vector<std::vector<Piece>> pieces(in.scans.size());
        for (size_t s = 0; s < in.scans.size() && par_ok; s++) {
            const JScan &js = in.scans[s];
            const uint32_t units = js.ncomp > 1 ? uint32_t(in.mcus_x * in.mcus_y)
                                                : uint32_t(in.comp[js.comp_idx[0]].real_bw * in.comp[js.comp_idx[0]].real_bh);
            if (in.restart_interval == 0) {
                if (js.has_marker) par_ok = false;   // marker bytes inside the data: leave it to the sequential kernel's libjpeg-like handling
                else pieces[s].push_back({js.data_off, js.data_len, 0u, units});
                continue;
            }
            // restart intervals: RSTm markers must come in order, one after every `restart_interval` MCUs, none missing
            const uint32_t ri = uint32_t(in.restart_interval), want = (units + ri - 1) / ri;
            size_t pos = js.data_off, end = js.data_off + js.data_len, start = pos;
            uint32_t idx = 0;
            bool ok = true;
            while (pos + 1 < end) {
                const void *f = memchr(d + pos, 0xFF, end - 1 - pos);
                if (!f) break;
                pos = size_t(static_cast<const uint8_t *>(f) - d);
                if (d[pos + 1] != 0x00) {
                    if (d[pos + 1] != 0xD0 + (idx & 7)) { ok = false; break; }
                    pieces[s].push_back({start, pos - start, idx * ri, std::min(ri, units - idx * ri)});
                    idx++;
                    if (idx >= want) { ok = false; break; }
                    pos += 2; start = pos;
                } else pos++;
            }
            if (ok && end > start && d[end - 1] == 0xFF) ok = false;
            if (ok) pieces[s].push_back({start, end - start, idx * ri, std::min(ri, units - idx * ri)});
            if (!ok || pieces[s].size() != want) par_ok = false;
        }
        if (par_ok) {
            for (size_t s = 0; s < in.scans.size(); s++) {
                const JScan &js = in.scans[s];
                const DecScan &ds = b->dscans[im.first_scan + s];
                for (const Piece &pc : pieces[s]) {
                    ParScan ps;
                    memset(&ps, 0, sizeof ps);
                    if (in.restart_interval == 0) { ps.bits_off = ds.bits_off; ps.bits_len = ds.bits_len; }
                    else {
                        ps.bits_off = uint32_t(b->bits_pool.size()); ps.bits_len = uint32_t(pc.len);
                        if (!b->bits_pool.append_aligned(d + pc.off, pc.len)) { csh_set_error("out of pinned host memory"); return CS_ERR_NO_DEVICE; }
                    }
                    ps.huff_set = ds.huff_set; ps.image = img_index; ps.ncomp = js.ncomp; ps.first_mcu = pc.first_mcu;
                    int m = 0;
                    for (int k = 0; k < js.ncomp; k++) {
                        const JComp &jc = in.comp[js.comp_idx[k]];
                        int nh = js.ncomp > 1 ? jc.h : 1, nv = js.ncomp > 1 ? jc.v : 1;
                        uint32_t nblocks = pc.nmcus * uint32_t(nh * nv);
                        for (int y = 0; y < nv; y++)
                            for (int x = 0; x < nh; x++, m++) {
                                if (m >= 10) break;
                                ps.comp_of[m] = js.comp_idx[k]; ps.by_of[m] = y; ps.bx_of[m] = x; ps.dct[m] = js.td[k]; ps.act[m] = js.ta[k];
                                ps.dc_base[m] = b->dc_total; ps.dc_per_mcu[m] = uint32_t(nh * nv); ps.dc_idx[m] = uint32_t(y * nh + x);
                            }
                        b->dc_total += nblocks;
                        ps.total_blocks += nblocks;
                    }
                    ps.nb_mcu = m;
                    uint32_t nsub = (ps.bits_len + CSH_SUBSEQ_BYTES - 1) / CSH_SUBSEQ_BYTES;
                    ps.sub_base = b->total_sub; ps.par_index = uint32_t(b->pscans.size());
                    b->total_sub += nsub;
                    b->max_sub = std::max(b->max_sub, nsub);
                    b->max_par_blocks = std::max(b->max_par_blocks, ps.total_blocks);
                    b->pscans.push_back(ps);
                }
            }
        }
        // progressive scans without restart markers: one wave per chain (k_decode_prog.hip); the scans are listed as ParScans
        // of kind 1 so that the unstuffing pre-pass covers them
        bool prog_ok = in.progressive && in.restart_interval == 0;
        for (size_t s = 0; s < in.scans.size(); s++) {
            const JScan &js = in.scans[s];
            if (js.has_marker || js.data_len >= (1u << 28) || !b->phset_fits[b->dscans[im.first_scan + s].huff_set]) prog_ok = false;
        }
        if (prog_ok) {
            // The scans of a progressive file, by what they are to the decoder (k_decode_par.hip, k_decode_prog.hip):
            //   DC first / AC first scans  self-synchronising like sequential scans -- cut into sub-sequences, speculated, relaxed, written
            //                              (CSH_PS_DC_FIRST / CSH_PS_AC_FIRST).  They carry most of a file's bits;
            //   DC refinement              one bit per block: unstuffed, then scattered (CSH_PS_DC_REFINE);
            //   AC refinement              a chain per component, one wave each (the bits between two symbols depend on the block's history:
            //                              a decoder that does not know its block cannot find the next symbol) -- CSH_PS_UNSTUFF + ProgChain.
            // CSH_PROG_PAR=0 (or a file beyond the parallel decoder's 24-bit block counts) leaves every scan to the chains, as before round 4.
            const char *pp = getenv("CSH_PROG_PAR");
            // The parallel kinds run side by side and in front of the refinement chains: sound only for a REGULAR progression -- every coefficient of every
            // component gets its first scan (Ah = 0) once and before any refinement, and each refinement takes over where the scan before it left the
            // coefficient (Ah = that scan's Al, Al = Ah - 1).  A damaged header can say otherwise (two first scans over one band: the later one wins in file
            // order, and two waves writing the same coefficient do not know which of them is later); libjpeg warns and decodes in file order: so do the chains.
            bool regular = true;
            {
                int state[CSH_MAX_COMPS][64];
                for (auto &row : state) for (int &v : row) v = -1;
                for (const JScan &js : in.scans)
                    for (int k = 0; k < js.ncomp && regular; k++) {
                        const int c = js.comp_idx[k];
                        if (c < 0 || c >= CSH_MAX_COMPS || js.Ss < 0 || js.Se > 63 || js.Ss > js.Se) { regular = false; break; }
                        for (int z = js.Ss; z <= js.Se; z++) {
                            if (js.Ah == 0 ? state[c][z] != -1 : (state[c][z] != js.Ah || js.Al != js.Ah - 1)) { regular = false; break; }
                            state[c][z] = js.Al;
                        }
                    }
            }
            const bool par_first = regular && !(pp && !strcmp(pp, "0")) && uint64_t(in.mcus_x) * uint64_t(in.mcus_y) * 10 < (1u << 24);
            for (size_t s = 0; s < in.scans.size(); s++) {
                const JScan &js = in.scans[s];
                DecScan &ds = b->dscans[im.first_scan + s];
                ParScan ps;
                memset(&ps, 0, sizeof ps);
                ps.bits_off = ds.bits_off; ps.bits_len = ds.bits_len; ps.huff_set = ds.huff_set; ps.image = img_index; ps.ncomp = ds.ncomp; ps.nb_mcu = 1;
                ps.Ss = js.Ss; ps.Se = js.Se; ps.Al = js.Al;
                ps.kind = !par_first ? CSH_PS_UNSTUFF : js.Ss == 0 ? (js.Ah == 0 ? CSH_PS_DC_FIRST : CSH_PS_DC_REFINE) : (js.Ah == 0 ? CSH_PS_AC_FIRST : CSH_PS_UNSTUFF);
                ps.sub_base = b->total_sub; ps.par_index = uint32_t(b->pscans.size());
                if (ps.kind != CSH_PS_UNSTUFF) {   // where the blocks of the scan lie (as for a sequential-mode scan: units are MCUs, or the blocks of its one component)
                    int m = 0;
                    for (int k = 0; k < js.ncomp; k++) {
                        const JComp &jc = in.comp[js.comp_idx[k]];
                        const int nh = js.ncomp > 1 ? jc.h : 1, nv = js.ncomp > 1 ? jc.v : 1;
                        const uint32_t units = js.ncomp > 1 ? uint32_t(in.mcus_x * in.mcus_y) : uint32_t(jc.real_bw * jc.real_bh);
                        const uint32_t nblocks = units * uint32_t(nh * nv);
                        for (int y = 0; y < nv; y++)
                            for (int x = 0; x < nh; x++, m++) {
                                if (m >= 10) break;
                                ps.comp_of[m] = js.comp_idx[k]; ps.by_of[m] = y; ps.bx_of[m] = x; ps.dct[m] = js.td[k]; ps.act[m] = js.ta[k];
                                ps.dc_base[m] = b->dc_total; ps.dc_per_mcu[m] = uint32_t(nh * nv); ps.dc_idx[m] = uint32_t(y * nh + x);
                            }
                        if (ps.kind == CSH_PS_DC_FIRST) b->dc_total += nblocks;
                        ps.total_blocks += nblocks;
                    }
                    ps.nb_mcu = m;
                    b->max_par_blocks = std::max(b->max_par_blocks, ps.total_blocks);
                    if (ps.kind != CSH_PS_DC_REFINE) {
                        const uint32_t nsub = (ps.bits_len + CSH_SUBSEQ_BYTES - 1) / CSH_SUBSEQ_BYTES;
                        b->total_sub += nsub;
                        b->max_sub = std::max(b->max_sub, nsub);
                    }
                }
                ds.par_index = int(b->pscans.size());
                b->pscans.push_back(ps);
            }
            // CSH_PROG_PAR=1: the first scans in parallel, refinement chains in one wave each (k_decode_prog.hip) as before k_decode_refine.hip
            const bool refine_split = par_first && !(pp && !strcmp(pp, "1"));
            for (int chain = 0; chain <= in.ncomp; chain++) {   // 0: DC scans; c + 1: AC scans of component c -- those the parallel decoder does not take
                ProgChain pc;
                memset(&pc, 0, sizeof pc);
                pc.image = img_index; pc.first = int(b->chain_scans.size());
                bool all_refine = chain != 0;
                for (size_t s = 0; s < in.scans.size(); s++) {
                    const JScan &js = in.scans[s];
                    const bool mine = chain == 0 ? js.Ss == 0 : (js.Ss != 0 && js.comp_idx[0] == chain - 1);
                    if (mine && b->pscans[size_t(b->dscans[im.first_scan + s].par_index)].kind == CSH_PS_UNSTUFF) {
                        b->chain_scans.push_back(im.first_scan + int(s)); pc.count++;
                        if (js.Ah == 0 || js.data_len >= (1u << 27)) all_refine = false;
                    }
                }
                if (pc.count && refine_split && all_refine) {   // parse + apply: room for a mask per block and a position per block and scan
                    const JComp &jc = in.comp[chain - 1];
                    const uint64_t nblocks = uint64_t(jc.real_bw) * uint64_t(jc.real_bh);
                    if (nblocks && uint64_t(b->refine_pos) + nblocks * uint64_t(pc.count) < (1ull << 32)) {
                        pc.refine = 1; pc.comp = chain - 1; pc.nblocks = uint32_t(nblocks);
                        pc.hist_off = b->refine_hist; pc.pos_off = b->refine_pos;
                        b->refine_hist += pc.nblocks; b->refine_pos += pc.nblocks * uint32_t(pc.count);
                        b->refine_max_blocks = std::max(b->refine_max_blocks, pc.nblocks);
                    }
                }
                if (pc.count) b->chains.push_back(pc);
            }
        }
        b->need_seq_init.push_back(par_ok ? 0u : (prog_ok ? 4u : 1u));

        // pixel work + planes
        if (!b->lossless)
            for (int c = 0; c < in.ncomp; c++) {
                PlaneWork w; w.image = img_index; w.comp = c;
                bool in_full = in.comp[c].h == in.hmax && in.comp[c].v == in.vmax;
                bool out_full = o.comp[c].h == o.hmax && o.comp[c].v == o.vmax;
                int in_kind = in_full ? 0 : (in.comp[c].v == in.vmax ? 2 : 1);    // 0 full, 1 h2v2, 2 h2v1
                int out_kind = out_full ? 0 : (o.comp[c].v == o.vmax ? 2 : 1);
                w.mode = (in_kind == 0 && out_kind == 0) ? 0 : 1 + 3 * in_kind + out_kind;
                if (resized) w.mode = 1 + out_kind;   // encoder side is fed full-resolution planes of the resized image (k_resize.hip)
                // the camera case (4:2:0 kept, no resize) goes through k_resample_fdct_420 and has no encoder-side plane
                const bool fused = w.mode == 5 && !resized && im.in[c].comp_w > 2 && im.in[c].real_bw == im.out[c].real_bw && im.in[c].real_bh == im.out[c].real_bh &&
                                   im.in[c].comp_w == im.out[c].comp_w && im.in[c].comp_h == im.out[c].comp_h && !getenv("CSH_NO_FUSED_420");
                if (fused) w.mode = 10;
                if (w.mode) {
                    im.plane_off[c] = plane_off;
                    plane_off += uint64_t(im.in[c].real_bw * 8) * uint64_t(im.in[c].real_bh * 8);
                    plane_off = (plane_off + 63u) & ~uint64_t(63);
                    im.splane_off[c] = im.plane_off[c];
                    if (resized) {   // full-resolution plane of the resized image, pitch = luma's padded width
                        JComp full; full.h = full.v = 1;
                        JpegInfo tmpj; tmpj.width = o.width; tmpj.height = o.height; tmpj.ncomp = 1; tmpj.comp[0] = full; jpeg_geometry(tmpj);
                        uint32_t dummy = 0; fill_geom(tmpj.comp[0], im.src[c], dummy);
                        im.splane_off[c] = plane_off;
                        plane_off += uint64_t(im.src[c].real_bw * 8) * uint64_t(im.src[c].real_bh * 8);
                        plane_off = (plane_off + 63u) & ~uint64_t(63);
                    }
                    im.oplane_off[c] = oplane_off;
                    const uint64_t osz = fused ? 0 : uint64_t(im.out[c].real_bw) * 8 * uint64_t(im.out[c].real_bh) * 8;   // <= 2^28 + edge blocks (plan_item)
                    oplane_off = (oplane_off + osz + 63u) & ~uint64_t(63);
                    b->max_quads = std::max(b->max_quads, uint32_t(osz / 4));
                }
                b->pwork.push_back(w);
            }

        if (resized) {
            ResizeWork rw;
            memset(&rw, 0, sizeof rw);
            rw.image = img_index; rw.nw = o.width; rw.nh = o.height;
            rw.in_kind = in.ncomp == 1 ? 0 : ((in.comp[1].h == in.hmax && in.comp[1].v == in.vmax) ? 0 : (in.comp[1].v == in.vmax ? 2 : 1));
            if (px) rw.in_kind = -1;   // the RGB of this image is not made from decoded planes: it is copied in below (csh_batch_create_from_pixels)
            const uint64_t src_bytes = uint64_t(in.width) * in.height * in.ncomp, dst_bytes = uint64_t(o.width) * o.height * in.ncomp;
            rw.rgb_src_off = b->rgb_bytes; b->rgb_bytes += (src_bytes + 63) & ~uint64_t(63);
            rw.rgb_dst_off = b->rgb_bytes; b->rgb_bytes += (dst_bytes + 63) & ~uint64_t(63);
            const uint64_t tmpn = uint64_t(o.height) * in.width * in.ncomp;
            rw.tmp_off = b->tmp_floats; b->tmp_floats += tmpn;
            const bool same = o.width == in.width && o.height == in.height;   // image-rs copies instead of resampling
            rw.vtap_base = uint32_t(b->rtaps.size()); csh_lanczos_axis(in.height, o.height, same, b->rtaps, b->rweights);
            rw.htap_base = uint32_t(b->rtaps.size()); csh_lanczos_axis(in.width, o.width, same, b->rtaps, b->rweights);
            b->max_src_px = std::max<uint32_t>(b->max_src_px, uint32_t(in.width) * in.height);
            b->max_tmp = std::max(b->max_tmp, tmpn);
            b->max_row_in = std::max(b->max_row_in, uint32_t(in.width) * uint32_t(in.ncomp)); b->max_out_w = std::max(b->max_out_w, uint32_t(o.width));
            b->max_nh = std::max(b->max_nh, uint32_t(o.height));
            b->max_dst = std::max(b->max_dst, dst_bytes);
            b->rwork.push_back(rw);
            if (b->webp) {
                csw::WebpImg wi;
                memset(&wi, 0, sizeof wi);
                wi.width = uint32_t(o.width); wi.height = uint32_t(o.height); wi.mbw = (wi.width + 15) / 16; wi.mbh = (wi.height + 15) / 16; wi.ncomp = uint32_t(in.ncomp);
                wi.rgb_off = rw.rgb_dst_off; wi.image = uint32_t(img_index);
                const uint64_t ly = uint64_t(wi.mbw) * wi.mbh * 256, lc = uint64_t(wi.mbw) * wi.mbh * 64;
                auto take = [&](uint64_t n) { uint64_t at = b->wwork_bytes; b->wwork_bytes += (n + 63) & ~uint64_t(63); return at; };
                wi.y_off = take(ly); wi.u_off = take(lc); wi.v_off = take(lc); wi.ry_off = take(ly); wi.ru_off = take(lc); wi.rv_off = take(lc);
                wi.lev_off = b->wlevels; b->wlevels += uint64_t(wi.mbw) * wi.mbh * csw::WEBP_MB_REC;
                b->wmax_luma = std::max<uint32_t>(b->wmax_luma, uint32_t(ly));
                b->wmax_mbh = std::max(b->wmax_mbh, wi.mbh);
                b->wimgs.push_back(wi);
            }
        }

        // output scans
        im.first_work = int(b->swork.size());
        if (!b->search) {
            int sb = in.ncomp == 3 ? script_base3 : script_base1;
            int ns = progressive ? (in.ncomp == 3 ? 10 : 6) : 1;
            std::vector<int> list;
            for (int s = 0; s < ns; s++) list.push_back(sb + s);
            add_works(it, im, img_index, list, inputs[n].length, o);
            for (int s = 0; s < ns; s++) b->img_list.push_back(uint32_t(im.first_work + s));
            b->img_list.resize(size_t(img_index + 1) * CSH_LIST_MAX, 0u);
            b->img_nlist.push_back(uint32_t(ns));
        } else {
            // stage 1 of the search: the DC scan and every candidate whose Al is fixed (oracle/jpeg_oracle.c cso_search_progression)
            csh_batch::SearchImg si;
            for (int &cw : si.cand_work) cw = -1;
            std::vector<int> list, cands;
            auto add = [&](int cand, int idx) { cands.push_back(cand); list.push_back(idx); };
            add(0, dc_scan_index(in.ncomp));
            add(1, cand_index(0, 1, 8, 0, 0)); add(2, cand_index(0, 9, 63, 0, 0));
            // luma at Al 1 and 2 (mozjpeg tries Al 3 only when Al 2 beat Al 1: candidates 9-11 are stage ST_1B)
            for (int Al = 0; Al < 2; Al++) { add(3 + 3 * Al, cand_index(0, 1, 63, Al + 1, Al)); add(4 + 3 * Al, cand_index(0, 1, 8, 0, Al + 1)); add(5 + 3 * Al, cand_index(0, 9, 63, 0, Al + 1)); }
            if (in.ncomp == 3) {
                add(26, cand_index(1, 1, 8, 0, 0)); add(27, cand_index(1, 9, 63, 0, 0)); add(28, cand_index(2, 1, 8, 0, 0)); add(29, cand_index(2, 9, 63, 0, 0));
                for (int Al = 0; Al < 2; Al++) {
                    add(30 + 6 * Al, cand_index(1, 1, 63, Al + 1, Al)); add(31 + 6 * Al, cand_index(2, 1, 63, Al + 1, Al));
                    add(32 + 6 * Al, cand_index(1, 1, 8, 0, Al + 1)); add(33 + 6 * Al, cand_index(1, 9, 63, 0, Al + 1));
                    add(34 + 6 * Al, cand_index(2, 1, 8, 0, Al + 1)); add(35 + 6 * Al, cand_index(2, 9, 63, 0, Al + 1));
                }
            }
            add_works(it, im, img_index, list, inputs[n].length, o);
            for (size_t k = 0; k < cands.size(); k++) si.cand_work[cands[k]] = im.first_work + int(k);
            si.ncand = in.ncomp == 3 ? 64 : 23;
            b->simg.push_back(si);
            b->img_list.resize(size_t(img_index + 1) * CSH_LIST_MAX, 0u);
            b->img_nlist.push_back(0u);
        }
        if (b->total_units > 0xFFFFFFF0ull) { it.code = CS_ERR_POOL_OVERFLOW; it.msg = "batch too large"; }

        // frame header (host-built): SOI, JFIF, [metadata], DQT, SOF
        JpegInfo hdr = o;
        if (b->lossless) memcpy(hdr.qt, in.qt, sizeof hdr.qt);           // coefficient transcode keeps the source tables
        else { memcpy(hdr.qt[0], qout_nat, 128); memcpy(hdr.qt[1], qout_nat, 128); }
        // metadata carry-over (host logic): APPn/COM when keep_metadata (compressor.rs:431); ICC profile segments follow
        // jpeg_preserve_icc = !--strip-icc (compressor.rs:425) independently of it
        std::vector<uint8_t> meta;
        for (size_t mo = 0; mo + 4 <= in.meta.size();) {
            size_t L = (size_t(in.meta[mo + 2]) << 8) | in.meta[mo + 3];
            bool is_icc = in.meta[mo + 1] == 0xE2 && L >= 14 && !memcmp(&in.meta[mo + 4], "ICC_PROFILE\0", 12);
            if (is_icc ? p->jpeg_preserve_icc : p->keep_metadata) meta.insert(meta.end(), in.meta.begin() + mo, in.meta.begin() + mo + 2 + L);
            mo += 2 + L;
        }
        it.meta_out = meta;
        std::vector<uint8_t> fh = build_frame_header(hdr, progressive, meta.empty() ? nullptr : &meta);
        b->hdr_off.push_back(uint32_t(b->hdr_pool.size()));
        b->hdr_pool.insert(b->hdr_pool.end(), fh.begin(), fh.end());

        it.image = img_index;
        b->imgs.push_back(im);
        b->raw_bytes_cap += (b->search ? 8 : 2) * inputs[n].length + (b->search ? 256 : 64) * 1024;   // the search's candidates are ten scripts' worth of scans (of the OUTPUT's size)
    }
    b->hdr_off.push_back(uint32_t(b->hdr_pool.size()));
    b->nimg = int(b->imgs.size());
    // stage boundary: everything made so far is stage 1 (without the search: all there is)
    b->stage_end(b->stage[0]);
    if (b->search) {
        static const int split[5] = {2, 8, 5, 12, 18};
        // the later stages, each contiguous: make(image, add) lists the stage's candidates of one image
        auto add_stage = [&](int sid, auto make) -> int {
            // one unused slot between the stages: each stage's exclusive scan of chunk sizes writes one entry past its slots
            b->nslots++;
            csh_batch::Stage &sg = b->stage[sid];
            b->stage_begin(sg);
            for (size_t n = 0; n < count; n++) {
                Item &it = b->items[n];
                if (it.image < 0) continue;
                ImgDesc &im = b->imgs[size_t(it.image)];
                csh_batch::SearchImg &si = b->simg[size_t(it.image)];
                std::vector<int> list, cands;
                auto add = [&](int cand, int idx) { cands.push_back(cand); list.push_back(idx); };
                make(im, add);
                const int first = int(b->swork.size());
                add_works(it, im, it.image, list, inputs[n].length, it.out);
                for (size_t k = 0; k < cands.size(); k++) si.cand_work[cands[k]] = first + int(k);
            }
            b->stage_end(sg);
            return 0;
        };
        // ST_1B: luma at Al 3 -- the refinement that brings it back to 2 and the two band scans
        add_stage(csh_batch::ST_1B, [&](const ImgDesc &, auto &add) { add(9, cand_index(0, 1, 63, 3, 2)); add(10, cand_index(0, 1, 8, 0, 3)); add(11, cand_index(0, 9, 63, 0, 3)); });
        // ST_2: the frequency-split candidates every search looks at: the whole band, the splits at 2 and at 5 (the split at 8 IS stage 1's
        // pair of band scans at the chosen Al: nothing is coded for it).  Their Al is the one stage 1 chooses: entered as 0, patched before the stage runs
        auto splits = [&](const ImgDesc &im, auto &add, int i0, int i1, bool whole) {
            if (whole) add(12, cand_index(0, 1, 63, 0, 0));
            for (int i = i0; i <= i1; i++) if (i != 1) { add(13 + 2 * i, cand_index(0, 1, split[i], 0, 0)); add(14 + 2 * i, cand_index(0, split[i] + 1, 63, 0, 0)); }
            if (im.ncomp == 3) {
                if (whole) { add(42, cand_index(1, 1, 63, 0, 0)); add(43, cand_index(2, 1, 63, 0, 0)); }
                for (int i = i0; i <= i1; i++) if (i != 1) {
                    add(44 + 4 * i, cand_index(1, 1, split[i], 0, 0)); add(45 + 4 * i, cand_index(1, split[i] + 1, 63, 0, 0));
                    add(46 + 4 * i, cand_index(2, 1, split[i], 0, 0)); add(47 + 4 * i, cand_index(2, split[i] + 1, 63, 0, 0));
                }
            }
        };
        add_stage(csh_batch::ST_2, [&](const ImgDesc &im, auto &add) { splits(im, add, 0, 2, true); });
        add_stage(csh_batch::ST_2B, [&](const ImgDesc &im, auto &add) { splits(im, add, 3, 3, false); });   // the split at 12: only if the split at 8 leads after the third
        add_stage(csh_batch::ST_2C, [&](const ImgDesc &im, auto &add) { splits(im, add, 4, 4, false); });   // the split at 18: only if the split at 12 leads after the fourth
        for (int c = 0; c < 3; c++) for (int i = -1; i < 5; i++) for (int Al = 1; Al <= (c ? 2 : 3); Al++) {   // the variants the Al patch may pick
            if (i < 0) cand_index(c, 1, 63, 0, Al); else { cand_index(c, 1, split[i], 0, Al); cand_index(c, split[i] + 1, 63, 0, Al); }
        }
        b->work_active.assign(b->swork.size(), 0);
        if (b->total_units > 0xFFFFFFF0ull) { csh_set_error("csh_batch_create: batch too large for the scan search's candidate lists (fewer files per batch)"); return CS_ERR_POOL_OVERFLOW; }
    }
    if (b->trellis) {
        // the trellis stage: per component one statistics scan in the output mode's entropy coder -- progressive: the component alone,
        // 1-63 at Al 0 (EOBRUN symbols included); sequential: a one-component sequential scan (DC and AC tables) -- coded for its
        // histograms only (mozjpeg jcmaster.c: the huff_opt pass in front of every trellis pass; oracle: cso_trellis_tables)
        b->nslots++;
        csh_batch::Stage &tg = b->tstage;
        b->stage_begin(tg);
        auto seq1_index = [&](int comp) -> int {
            const std::array<int, 5> key = {comp, 0, 63, -1, -1};
            auto f = b->cand_script.find(key);
            if (f != b->cand_script.end()) return f->second;
            EncScan e;
            memset(&e, 0, sizeof e);
            const int id = comp ? 1 : 0;
            e.ncomp = 1; e.comp[0] = comp; e.Ss = 0; e.Se = 63; e.sequential = 1;
            e.ntables = 2; e.dht_id[0] = id; e.dht_id[1] = 0x10 | id; e.dc_tbl[0] = 0; e.ac_tbl[0] = 1; e.sos_tdta[0] = (id << 4) | id;
            b->script.push_back(e);
            return b->cand_script[key] = int(b->script.size()) - 1;
        };
        for (size_t n = 0; n < count; n++) {
            Item &it = b->items[n];
            if (it.image < 0) continue;
            ImgDesc &im = b->imgs[size_t(it.image)];
            std::vector<int> list;
            for (int c = 0; c < im.ncomp; c++) list.push_back(progressive ? cand_index(c, 1, 63, 0, 0) : seq1_index(c));
            const size_t first = b->swork.size();
            add_works(it, im, it.image, list, inputs[n].length, it.out, true);
            for (int c = 0; c < im.ncomp; c++) {
                const ScanWork &sw = b->swork[first + size_t(c)];
                TrellisWork tw;
                memset(&tw, 0, sizeof tw);
                tw.image = it.image; tw.comp = c;
                tw.table_ac = sw.table_base + (progressive ? 0u : 1u);
                tw.table_dc = progressive ? -1 : int32_t(sw.table_base);
                tw.nunits = sw.nunits; tw.unit_base = b->t_units;
                b->t_units += sw.nunits;
                for (uint32_t j = 0; j < (sw.nunits + CSH_TR_WG - 1) / CSH_TR_WG; j++) b->tchunks.push_back(TrellisChunk{uint32_t(b->twork.size()), j});
                b->t_max_rows = std::max<uint32_t>(b->t_max_rows, uint32_t((im.out[c].real_bh + im.out[c].v - 1) / im.out[c].v));
                tw.nzset = 0xFFFFFFFFu;
                if (progressive) {
                    const int si = b->nzset_of[size_t(it.image) * CSH_MAX_COMPS + size_t(c)];
                    b->nzsets[size_t(si)].cnt_base = tw.unit_base;   // the statistics list's builder counts every block's entries
                    if (b->nz_once) tw.nzset = uint32_t(si);
                }
                b->twork.push_back(tw);
            }
        }
        b->stage_end(tg);
        {   // the DC walks, longest first (a wave's 64 lanes then walk rows of a length)
            const char *ts = getenv("CSH_TR_SORT");
            b->t_sort = progressive && !(ts && !strcmp(ts, "0"));
            std::vector<std::pair<uint32_t, uint32_t>> rows;
            for (size_t wi = 0; wi < b->twork.size(); wi++) {
                const CompGeom &g = b->imgs[size_t(b->twork[wi].image)].out[b->twork[wi].comp];
                for (int r = 0; r < (g.real_bh + g.v - 1) / g.v; r++) rows.push_back({uint32_t(g.real_bw * g.v), uint32_t(wi << 16) | uint32_t(r)});
            }
            std::stable_sort(rows.begin(), rows.end(), [](const auto &a, const auto &b2) { return a.first > b2.first; });
            for (const auto &r : rows) b->trows.push_back(r.second);
        }
        if (b->total_units > 0xFFFFFFF0ull) { csh_set_error("csh_batch_create: batch too large (fewer files per batch)"); return CS_ERR_POOL_OVERFLOW; }
    }
    // pool layout: [all decoded tiles][all re-quantised tiles]; only the first part must start at zero for the decoder
    if (!b->lossless)
        for (ImgDesc &im : b->imgs) for (int c = 0; c < im.ncomp; c++) im.out[c].tile_base += b->ntiles_in;
    for (size_t i = 0; i < b->plans.size(); i++) b->plans[i].tile_base = b->imgs[size_t(b->plan_image[i])].out[b->plan_comp[i]].tile_base;
    for (size_t i = 0; i < b->nzsets.size(); i++) b->nzsets[i].tile_base = b->imgs[size_t(b->nzset_image[i])].out[b->nzset_comp[i]].tile_base;
    for (ParScan &ps : b->pscans)   // table selectors: slot numbers of the table-set form the batch uses
        for (int m = 0; m < ps.nb_mcu && m < 10; m++) {
            int dcs = ps.dct[m] & 3, acs = 4 + (ps.act[m] & 3);
            if (b->use4) { dcs = std::max<int>(0, b->slot4[ps.huff_set][dcs]); acs = std::max<int>(0, b->slot4[ps.huff_set][acs]); }
            ps.sel |= uint64_t(dcs | (acs << 3)) << (6 * m);
        }
    {   // the scans of the refinement chains as units of k_refine_parse, scan-major: a unit's predecessor has a smaller number
        std::vector<int> last(b->chains.size(), -1);
        for (int s = 0;; s++) {
            bool any = false;
            for (size_t c = 0; c < b->chains.size(); c++)
                if (b->chains[c].refine && b->chains[c].count > s) {
                    RefineUnit u; u.chain = int(c); u.s = s; u.prev = last[c];
                    last[c] = int(b->refine_units.size());
                    b->refine_units.push_back(u);
                    any = true;
                }
            if (!any) break;
        }
    }
    b->ntiles = b->ntiles_in + b->ntiles_out;
    b->plane_bytes = plane_off;
    b->oplane_bytes = oplane_off;
    b->out_cap = b->raw_bytes_cap;
    layout_token_pool(b.get());

    lap("descriptors");
    // upload what never changes between runs
    hipStream_t st = b->stream;
    if (b->nimg) {
        b->bits_pool.flush_copies();
        lap("copy_in");
        if (b->d_bits.alloc(b->bits_pool.size()) || (b->bits_pool.size() && hipMemcpyAsync(b->d_bits.p, b->bits_pool.p, b->bits_pool.size(), hipMemcpyHostToDevice, st) != hipSuccess) ||
            b->d_imgs.upload(b->imgs, st) || b->d_dscans.upload(b->dscans, st) || b->d_chains.upload(b->chains, st) || b->d_chain_scans.upload(b->chain_scans, st) ||
            b->d_hsets.upload(b->hsets, st) || b->d_phsets.upload(b->phsets, st) || (b->use4 && b->d_phsets4.upload(b->phsets4, st)) || b->d_quants.upload(b->quants, st) || b->d_pwork.upload(b->pwork, st) ||
            b->d_script.upload(b->script, st) || b->d_swork.upload(b->swork, st) || b->d_slot_work.alloc(size_t(b->nslots) + 1) || b->d_slots.alloc(size_t(b->nslots) + 1) || b->d_plans.upload(b->plans, st) || b->d_echunks.upload(b->echunks, st) || b->d_hdr.upload(b->hdr_pool, st) ||
            b->d_hdr_off.upload(b->hdr_off, st) || b->d_pscans.upload(b->pscans, st) || b->d_rwork.upload(b->rwork, st) || b->d_rtaps.upload(b->rtaps, st) ||
            b->d_rweights.upload(b->rweights, st) || b->d_rgb.alloc(b->rgb_bytes + 64) || b->d_rtmp.alloc((resize_is_fused(b->max_row_in) ? 0 : b->tmp_floats) + 16) || b->d_need_seq_init.upload(b->need_seq_init, st))
            return CS_ERR_NO_DEVICE;
        {
            size_t nchunks = b->bits_pool.size() / 64 + 1, nst = size_t(b->total_sub) + b->pscans.size() + 1;
            if (b->d_clean.alloc(b->bits_pool.size() + 64) || b->d_unstuff_cnt.alloc(nchunks + 1) || b->d_unstuff_off.alloc(nchunks + 2) ||
                b->d_pstate.alloc(nst) || b->d_relax_list[0].alloc(nst) || b->d_relax_list[1].alloc(nst) || b->d_relax_cnt.alloc(512) || b->d_scan_pending.alloc(b->pscans.size() + 1) || b->d_cut_block.alloc(b->pscans.size() + 1) || b->d_claim.alloc(size_t(b->total_sub) + 1) || b->d_hyp.alloc((size_t(b->total_sub) + 1) * 10) ||
                b->d_nblk.alloc(size_t(b->total_sub) + 1) || b->d_blk_off.alloc(size_t(b->total_sub) + 2) || b->d_need_seq.alloc(b->nimg + 1) ||
                b->d_dcdiff.alloc(size_t(b->dc_total) + 1) || b->d_dc_off.alloc(size_t(b->dc_total) + 2) ||
                b->d_refine_hist.alloc(size_t(b->refine_hist) + 1) || b->d_refine_pos.alloc(size_t(b->refine_pos) + 1) || b->d_refine_units.upload(b->refine_units, st) ||
                b->d_refine_prog.alloc(b->refine_units.size() + 1))
                return CS_ERR_NO_DEVICE;
        }
        if (b->d_coef.alloc(size_t(b->ntiles) * CSH_TILE_I16) || b->d_planes.alloc(b->plane_bytes + 64) || b->d_oplanes.alloc(b->oplane_bytes + 64) ||
            b->d_symbits.alloc(b->total_words + 1) || b->d_eobbits.alloc(b->total_words + 1) ||
            b->d_long_runs.alloc(2 * (b->total_units / 512 + b->swork.size() + 16)) || b->d_long_cnt.alloc(4) || b->d_tail.alloc(b->total_units + 1) || b->d_eobrun.alloc(b->total_units + 1) || b->d_corr.alloc(b->total_corr + 1) ||
            b->d_tok_off.alloc(4 * size_t(b->nslots) + 4) || b->d_chunk_ntok.alloc(4 * size_t(b->nslots) + 4) || b->d_slot_hist.alloc(size_t(b->hist_rows) * 256 + 256) ||
            b->d_slot_raw.alloc(size_t(b->nslots) + 1) || b->d_img_list.upload(b->img_list, st) || b->d_img_nlist.upload(b->img_nlist, st) || b->d_scan_cost.alloc(b->swork.size() + 1) || b->d_slot_eobh.alloc(16 * size_t(b->nslots) + 16) || b->d_chunk_bits.alloc(size_t(b->nslots) + 1) || b->d_chunk_off.alloc(size_t(b->nslots) + 2) || b->d_tok_cursor.alloc(b->region_est.size() + 1) || b->d_regions.upload(b->regions, st) ||
            b->d_tables.alloc(b->ntables) || b->d_scan_pad.alloc(b->swork.size() + 1) ||
            b->d_nzlists.upload(b->nzlists, st) || b->d_nzsets.upload(b->nzsets, st) || b->d_nzchunks.upload(b->nzchunks, st) || b->d_list_slots.alloc(size_t(b->nlist_slots) + 1) || b->d_tok_slots.alloc(size_t(b->ntok_slots) + 1) ||
            b->d_nz_cursor.alloc(b->nzlists.size() + 1) || b->d_nz_chunk_off.alloc(size_t(b->nz_nrec) + 1) || b->d_nz_chunk_cnt.alloc(size_t(b->nz_nrec) + 1) ||
            b->d_scan_raw_off.alloc(b->swork.size() + 2) || b->d_img_size.alloc(b->nimg + 1) || b->d_img_size_pad.alloc(b->nimg + 1) ||
            b->d_img_off.alloc(b->nimg + 2) || b->d_status.alloc(b->nimg) || b->d_overflow.alloc(4))
            return CS_ERR_NO_DEVICE;
        // the slots of every work item, written where they are used (the host counted them: add_works)
        if (hipMemsetAsync(b->d_slots.p, 0, (size_t(b->nslots) + 1) * sizeof(SlotRec), st) != hipSuccess ||
            hipMemsetAsync(b->d_slot_work.p, 0, (size_t(b->nslots) + 1) * sizeof(uint32_t), st) != hipSuccess) { csh_set_error("hipMemsetAsync failed"); return CS_ERR_NO_DEVICE; }
        launch_make_slots(st, b->d_swork.p, uint32_t(b->swork.size()), b->d_script.p, b->d_nzlists.p, b->d_slots.p, b->d_slot_work.p, b->d_list_slots.p, b->d_tok_slots.p);
        if (b->trellis && (b->d_trows.upload(b->trows, st) || (b->t_sort && (b->d_tperm.alloc(size_t(b->t_units) + 1) || b->d_tblk_cnt.alloc(size_t(b->t_units) + 1) || (b->nz_once && b->d_tblk_off.alloc(size_t(b->t_units) + 1)))))) return CS_ERR_NO_DEVICE;
        if (b->trellis && (b->d_twork.upload(b->twork, st) || b->d_tchunks.upload(b->tchunks, st) || b->d_tlambda.alloc(size_t(b->t_units) + 1) || b->d_tdcbt.alloc(size_t(b->t_units) + 1) ||
                           b->d_tspill.alloc(trellis_spill_words()) || b->d_dct_raw.alloc(size_t(b->ntiles_out) * CSH_TILE_I16)))
            return CS_ERR_NO_DEVICE;
        for (size_t n = 0; px && n < count; n++) {
            const Item &it = b->items[n];
            if (it.image < 0) continue;
            const ResizeWork &rw = b->rwork[size_t(it.image)];   // every image of such a batch has its resize work item, in image order
            if (rw.image != it.image) { csh_set_error("internal: resize work out of order"); return CS_ERR_NO_DEVICE; }
            if (hipMemcpyAsync(b->d_rgb.p + rw.rgb_src_off, px[n].device_pixels, size_t(px[n].width) * px[n].height * px[n].channels, hipMemcpyDeviceToDevice, st) != hipSuccess) { csh_set_error("pixel copy failed"); return CS_ERR_NO_DEVICE; }
        }
        lap("alloc+enqueue");
        if (hipStreamSynchronize(st) != hipSuccess) { csh_set_error("upload failed"); return CS_ERR_NO_DEVICE; }
        lap("upload_wait");
    }
    if (trace) fprintf(stderr, "[csh] create %zu files:%s\n", count, laps.c_str());
    *out = b.release();
    return 0;
}

// Pixels in, JPEG out (the back half of convert_in_memory to JPEG): the batch object derives every descriptor from a parsed JPEG, so a
// pixel source presents itself as one -- a baseline 4:4:4 (or grey) file of its size whose every block is empty (two bits per block:
// DC difference 0, end of block) -- and its RGB is copied over what those planes would have given, in front of the resize branch.
// The decode of the stand-in costs a few bits per block and one IDCT over zeros; everything behind the RGB is the resize path as is.
static std::vector<uint8_t> standin_jpeg(uint32_t w, uint32_t h, uint32_t nc) {
    std::vector<uint8_t> f = {0xFF, 0xD8, 0xFF, 0xDB, 0x00, 0x43, 0x00};
    f.insert(f.end(), 64, 1);                                                                    // DQT 0: all ones
    const uint8_t sof[] = {0xFF, 0xC0, 0x00, uint8_t(8 + 3 * nc), 8, uint8_t(h >> 8), uint8_t(h), uint8_t(w >> 8), uint8_t(w), uint8_t(nc)};
    f.insert(f.end(), sof, sof + sizeof sof);
    for (uint32_t c = 0; c < nc; c++) { f.push_back(uint8_t(c + 1)); f.push_back(0x11); f.push_back(0); }
    for (int cls = 0; cls < 2; cls++) {                                                          // DHT: one 1-bit code, symbol 0 (DC category 0 / AC end of block)
        const uint8_t dht[] = {0xFF, 0xC4, 0x00, 0x14, uint8_t(cls << 4), 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x00};
        f.insert(f.end(), dht, dht + sizeof dht);
    }
    const uint8_t sos[] = {0xFF, 0xDA, 0x00, uint8_t(6 + 2 * nc), uint8_t(nc)};
    f.insert(f.end(), sos, sos + sizeof sos);
    for (uint32_t c = 0; c < nc; c++) { f.push_back(uint8_t(c + 1)); f.push_back(0x00); }
    f.push_back(0); f.push_back(63); f.push_back(0);
    const uint64_t bits = uint64_t((w + 7) / 8) * ((h + 7) / 8) * nc * 2;
    f.insert(f.end(), size_t(bits / 8), 0x00);
    if (bits % 8) f.push_back(uint8_t(0xFF >> (bits % 8)));                                      // the last byte is padded with one-bits
    f.push_back(0xFF); f.push_back(0xD9);
    return f;
}
extern "C" int csh_batch_create_from_pixels(const csp_pixels *sources, size_t count, const CCSParameters *p, int device, csh_batch **out) {
    *out = nullptr;
    std::vector<std::vector<uint8_t>> files(count);
    std::vector<CByteArray> in(count);
    for (size_t i = 0; i < count; i++) {
        const csp_pixels &s = sources[i];
        if (!s.device_pixels || !s.width || !s.height || s.width > 65535 || s.height > 65535 || (s.channels != 1 && s.channels != 3)) files[i] = {'?'};   // answered per file: unknown type
        else files[i] = standin_jpeg(s.width, s.height, s.channels);
        in[i].data = files[i].data(); in[i].length = files[i].size();
    }
    return batch_create(in.data(), count, p, device, false, out, false, sources);
}

// pixels in, (resized) pixels out: the resize branch alone, for WebP -> PNG with a size
extern "C" int csh_batch_create_from_pixels_rgb(const csp_pixels *sources, size_t count, const CCSParameters *p, int device, csh_batch **out) {
    *out = nullptr;
    std::vector<std::vector<uint8_t>> files(count);
    std::vector<CByteArray> in(count);
    for (size_t i = 0; i < count; i++) {
        const csp_pixels &s = sources[i];
        if (!s.device_pixels || !s.width || !s.height || s.width > 65535 || s.height > 65535 || (s.channels != 1 && s.channels != 3)) files[i] = {'?'};
        else files[i] = standin_jpeg(s.width, s.height, s.channels);
        in[i].data = files[i].data(); in[i].length = files[i].size();
    }
    return batch_create(in.data(), count, p, device, false, out, true, sources);
}
extern "C" int csh_batch_create_webp_from_pixels(const csp_pixels *sources, size_t count, const CCSParameters *p, int device, csh_batch **out) {
    *out = nullptr;
    std::vector<std::vector<uint8_t>> files(count);
    std::vector<CByteArray> in(count);
    for (size_t i = 0; i < count; i++) {
        const csp_pixels &s = sources[i];
        if (!s.device_pixels || !s.width || !s.height || s.width > 16383 || s.height > 16383 || (s.channels != 1 && s.channels != 3)) files[i] = {'?'};
        else files[i] = standin_jpeg(s.width, s.height, s.channels);
        in[i].data = files[i].data(); in[i].length = files[i].size();
    }
    return batch_create(in.data(), count, p, device, true, out, false, sources);
}

extern "C" void csh_batch_destroy(csh_batch *b) { delete b; }

// kernel timing slots (csh_timing.kernel_ms); names via csh_kernel_name()
static const char *const kKernelNames[CSH_NKERNELS] = {
    "memset_coef", "unstuff", "k_dec_spec", "k_dec_relax0", "k_dec_relax1_4", "k_dec_write", "k_dc_scatter", "k_refine_chains", "k_decode_prog+seq",
    "k_idct_plane", "resize", "k_xform_direct", "k_resample+k_plane_fdct", "k_fix_dummy", "memset_enc", "trellis_stats", "k_trellis_ac", "k_trellis_dc",
    "k_nzlist", "k_tokens", "k_list_stats", "k_ac_runs", "k_gen_tables", "k_chunk_sizes", "scan_chunk_bits", "scan_layout", "k_pack", "k_list_pack",
    "k_ff_count", "scan_search_stage2", "k_layout", "scan_images", "k_emit", "", "", ""};
// a WebP batch (csh_batch_create_webp) leaves the JPEG path behind the resize slot: its next three slots are these
static const char *const kWebpTailNames[3] = {"k_webp_yuv", "k_vp8_analyse+segments+loop", "k_webp_hdr+decisions+bool+assemble"};

// the trellis slots (statistics scan = k_tokens without tokens + k_ac_runs + k_gen_tables; the two k_trellis kernels + k_fix_dummy) count
// as phase 1: they are the quantiser; zero unless CSH_PROFILE=mozjpeg
static const int kKernelPhase[CSH_NKERNELS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 1, 1, 1, 2, 2, 2, 2, 3, 4, 4, 4, 5, 5, 6, 7, 6, 6, 6, 7, 7, 7};
extern "C" const char *csh_kernel_name(int i) { return (i >= 0 && i < CSH_NKERNELS) ? kKernelNames[i] : ""; }
extern "C" const char *csh_kernel_name_webp(int i) { return (i >= 11 && i < 14) ? kWebpTailNames[i - 11] : csh_kernel_name(i); }

// the WebP tail of a run: RGB (resize branch) -> YUV 4:2:0 -> macroblocks -> tokens; files land in the batch's output pool at
// fixed offsets (capacity per macroblock grows on overflow, like the JPEG pools)
static int run_webp(csh_batch *b, csh_timing *t, hipEvent_t *ev, int slot) {
    hipStream_t st = b->stream;
    const int nimg = int(b->wimgs.size());
    uint64_t out_bytes = 0;
    std::vector<uint64_t> off(size_t(b->nimg) + 1, 0);
    for (auto &wi : b->wimgs) {
        const uint64_t cap = 4096 + uint64_t(wi.mbw) * wi.mbh * (b->webp_mb_bytes + 2);
        wi.out_cap = uint32_t(std::min<uint64_t>(cap, 0xFFFFFF00u)); wi.out_off = out_bytes;
        off[wi.image] = out_bytes;
        out_bytes += (wi.out_cap + 63) & ~uint64_t(63);
        const int q = int(b->params.webp_quality);
        wi.quality = q < 0 ? 0 : q > 100 ? 100 : q;
    }
    off[b->nimg] = out_bytes;
    if (b->d_out.n < out_bytes + 64 && b->d_out.alloc(out_bytes + 64)) return -1;
    if ((b->d_wscratch.n < out_bytes + 64 && b->d_wscratch.alloc(out_bytes + 64)) || (b->d_wpart.n < size_t(b->nimg) * 9 + 9 && b->d_wpart.alloc(size_t(b->nimg) * 9 + 9)) ||
        (b->d_wstats.n < size_t(b->nimg) * 2112 + 8 && (b->d_wstats.alloc(size_t(b->nimg) * 2112 + 8) || b->d_wprobs.alloc(size_t(b->nimg) * 1056 + 8) || b->d_wupdate.alloc(size_t(b->nimg) * 1056 + 8))))
        return -1;
    if (b->d_wstats.zero(st)) return -1;
    if (b->d_wimgs.upload(b->wimgs, st) || (b->d_wwork.n < b->wwork_bytes + 64 && b->d_wwork.alloc(b->wwork_bytes + 64)) ||
        (b->d_wlevels.n < b->wlevels + 64 && b->d_wlevels.alloc(b->wlevels + 64)))
        return -1;
    CSH_CHECK(hipMemcpyAsync(b->d_img_off.p, off.data(), off.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    if (b->d_img_size.zero(st)) return -1;
    csw::launch_webp_yuv(st, b->d_wimgs.p, nimg, b->wmax_luma, b->d_rgb.p, b->d_wwork.p);
    CSH_CHECK(hipEventRecord(ev[++slot], st));
    const int mid = ++slot;
    if (csw::launch_webp_encode(st, b->wimgs.data(), nimg, b->d_wimgs.p, b->d_wwork.p, b->d_wlevels.p, b->d_wscratch.p, b->d_wpart.p, b->d_out.p, b->d_img_size.p, b->d_status.p, ev[mid])) return -1;
    CSH_CHECK(hipEventRecord(ev[++slot], st));
    CSH_CHECK(hipStreamSynchronize(st));
    CSH_CHECK(hipGetLastError());
    if (t) {
        for (int i = 0; i < slot; i++) CSH_CHECK(hipEventElapsedTime(&t->kernel_ms[i], ev[i], ev[i + 1]));
        CSH_CHECK(hipEventElapsedTime(&t->total_ms, ev[0], ev[slot]));
        t->n_images = uint32_t(b->nimg);
    }
    for (int i = 0; i <= CSH_NKERNELS; i++) (void)hipEventDestroy(ev[i]);
    return 0;
}

// csh_batch_create_pixels: nothing behind the resize branch
static int run_rgb_only(csh_batch *b, csh_timing *t, hipEvent_t *ev, int slot) {
    hipStream_t st = b->stream;
    CSH_CHECK(hipEventRecord(ev[++slot], st));
    CSH_CHECK(hipStreamSynchronize(st));
    CSH_CHECK(hipGetLastError());
    if (t) {
        for (int i = 0; i < slot; i++) CSH_CHECK(hipEventElapsedTime(&t->kernel_ms[i], ev[i], ev[i + 1]));
        CSH_CHECK(hipEventElapsedTime(&t->total_ms, ev[0], ev[slot]));
        t->n_images = uint32_t(b->nimg);
    }
    for (int i = 0; i <= CSH_NKERNELS; i++) (void)hipEventDestroy(ev[i]);
    return 0;
}

// The host side of mozjpeg's scan search (jcmaster.c select_scans [UPSTREAM-RECALL]; the statement the oracle is pinned with:
// oracle/jpeg_oracle.c cso_search_progression).  The device has coded a stage's candidate scans; their sizes (DHT + SOS + stuffed data)
// come back and the decisions are replayed per image in mozjpeg's own order -- which is also what decides whether an image needs a
// conditional stage at all:
//   after ST_1   luma Al 0, 1, 2 in turn (stop at the first that is not cheaper); chroma Al 0, 1, 2 likewise.  Al 2 cheaper than Al 1:
//                the image wants luma at Al 3 tried (ST_1B)
//   after ST_1B  luma Al 3.  Then the Al of the frequency-split candidates is known: their work items and token plans are patched
//   after ST_2   whole band, split at 2, split at 8 (= stage 1's band pair at the chosen Al: search_work), [stop if the whole band still
//                leads], split at 5, [stop unless the split at 8 leads]: luma and chroma apart.  Not stopped: the split at 12 (ST_2B)
//   after ST_2B  split at 12, [stop unless it leads].  Not stopped: the split at 18 (ST_2C)
//   after ST_2C  split at 18.  Then every file's list of scans.
// Candidate numbering: cso_search_progression's.

enum { kLumaSplit0 = 12, kNLuma = 23, kChromaBase = 26, kChromaSplit0 = 42 };
// the work item that holds candidate `cand` of an image: its own, or -- the split at 8 -- stage 1's band scans at the chosen Al
static int search_work(const csh_batch::SearchImg &si, int cand) {
    if (cand == kLumaSplit0 + 3 || cand == kLumaSplit0 + 4) return si.cand_work[1 + 3 * si.Al_luma + (cand - (kLumaSplit0 + 3))];
    if (cand >= kChromaSplit0 + 6 && cand <= kChromaSplit0 + 9) return si.cand_work[kChromaBase + 6 * si.Al_chroma + (cand - (kChromaSplit0 + 6))];
    return si.cand_work[cand];
}
static int search_costs(csh_batch *b, AsmCtx &a, int stage) {
    hipStream_t st = b->stream;
    const csh_batch::Stage &sg = b->stage[stage];
    a.work0 = int(sg.work0); a.nwork_run = int(sg.nwork);
    launch_scan_cost(st, a);
    b->h_cost.resize(b->swork.size());
    CSH_CHECK(hipMemcpyAsync(b->h_cost.data() + sg.work0, b->d_scan_cost.p + sg.work0, size_t(sg.nwork) * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    CSH_CHECK(hipStreamSynchronize(st));
    return 0;
}
// marks the work items of `stage` of every image for which want(image) holds; returns how many images that is
template <class F>
static uint32_t search_gate(csh_batch *b, int stage, F want) {
    const csh_batch::Stage &sg = b->stage[stage];
    b->work_active.resize(b->swork.size());
    uint32_t nimg = 0;
    std::vector<char> on(size_t(b->nimg), 0);
    for (int i = 0; i < b->nimg; i++) { const int m = want(i); on[size_t(i)] = char(m); if (m) nimg++; }
    for (uint32_t wi = sg.work0; wi < sg.work0 + sg.nwork; wi++) {
        const ScanWork &w = b->swork[wi];
        const int comp = b->script[size_t(w.scan)].comp[0];
        b->work_active[wi] = uint8_t((on[size_t(w.image)] & (comp == 0 ? 1 : 2)) ? 1 : 0);   // want: bit 0 luma, bit 1 chroma
    }
    return nimg;
}
static int search_decide(csh_batch *b, int stage) {
    hipStream_t st = b->stream;
    for (int i = 0; i < b->nimg; i++) {
        csh_batch::SearchImg &si = b->simg[size_t(i)];
        const ImgDesc &im = b->imgs[size_t(i)];
        auto size = [&](int cand) -> uint64_t { return b->h_cost[size_t(search_work(si, cand))]; };
        auto luma_split = [&](int idx) { return idx == 0 ? size(kLumaSplit0) : size(kLumaSplit0 + 2 * idx - 1) + size(kLumaSplit0 + 2 * idx); };
        auto chroma_split = [&](int idx) {
            if (idx == 0) return size(kChromaSplit0) + size(kChromaSplit0 + 1);
            uint64_t cost = 0;
            for (int k = 2; k <= 5; k++) cost += size(kChromaSplit0 + 4 * (idx - 1) + k);
            return cost;
        };
        // one step of the split loop (jcmaster.c): returns true when the search goes on to idx + 1
        auto split_step = [&](int idx, uint64_t cost, uint64_t &best, int &choice) {
            if (idx == 0) { best = cost; choice = 0; return true; }
            if (cost < best) { best = cost; choice = idx; }
            return !((idx == 2 && choice == 0) || (idx == 3 && choice != 2) || (idx == 4 && choice != 4) || idx == 5);
        };
        if (stage == csh_batch::ST_1) {
            si.Al_luma = 0; si.Al_chroma = 0; si.luma_on = true; si.chroma_on = false;
            for (int Al = 0; Al <= 2 && si.luma_on; Al++) {   // candidates 1+3Al, 2+3Al: the two band scans at Al; 3+3k: the refinements that bring it back to 0
                uint64_t cost = size(1 + 3 * Al) + size(2 + 3 * Al);
                for (int k = 0; k < Al; k++) cost += size(3 + 3 * k);
                if (Al == 0 || cost < si.best_luma) { si.best_luma = cost; si.Al_luma = Al; } else si.luma_on = false;
            }
            if (im.ncomp == 3)
                for (int Al = 0; Al <= 2; Al++) {
                    uint64_t cost = 0;
                    for (int k = 0; k < 4; k++) cost += size(kChromaBase + 6 * Al + k);
                    for (int k = 0; k < Al; k++) cost += size(kChromaBase + 4 + 6 * k) + size(kChromaBase + 5 + 6 * k);
                    if (Al == 0 || cost < si.best_chroma) { si.best_chroma = cost; si.Al_chroma = Al; } else break;
                }
        } else if (stage == csh_batch::ST_1B) {
            if (si.luma_on) {
                const uint64_t cost = size(10) + size(11) + size(3) + size(6) + size(9);
                if (cost < si.best_luma) { si.best_luma = cost; si.Al_luma = 3; }
                si.luma_on = false;
            }
        } else if (stage == csh_batch::ST_2) {
            si.luma_on = true;
            for (int idx = 0; idx <= 3 && si.luma_on; idx++) si.luma_on = split_step(idx, luma_split(idx), si.best_luma, si.split_luma);
            si.chroma_on = im.ncomp == 3;
            for (int idx = 0; idx <= 3 && si.chroma_on; idx++) si.chroma_on = split_step(idx, chroma_split(idx), si.best_chroma, si.split_chroma);
        } else {
            const int idx = stage == csh_batch::ST_2B ? 4 : 5;
            if (si.luma_on) si.luma_on = split_step(idx, luma_split(idx), si.best_luma, si.split_luma);
            if (si.chroma_on) si.chroma_on = split_step(idx, chroma_split(idx), si.best_chroma, si.split_chroma);
        }
    }
    if (stage == csh_batch::ST_1 || stage == csh_batch::ST_1B) {
        // once no image waits for ST_1B: the frequency-split stages are coded at the chosen Al -- their work items' scans, and the Al in their token plans
        bool pending = false;
        for (int i = 0; i < b->nimg && stage == csh_batch::ST_1; i++) pending = pending || b->simg[size_t(i)].luma_on;
        if (pending) return 0;
        for (int sid : {int(csh_batch::ST_2), int(csh_batch::ST_2B), int(csh_batch::ST_2C)}) {
            const csh_batch::Stage &sg = b->stage[sid];
            for (uint32_t wi = sg.work0; wi < sg.work0 + sg.nwork; wi++) {
                ScanWork &w = b->swork[wi];
                const EncScan e = b->script[size_t(w.scan)];
                const csh_batch::SearchImg &si = b->simg[size_t(w.image)];
                const int Al = e.comp[0] == 0 ? si.Al_luma : si.Al_chroma;
                w.scan = b->cand_script.at({e.comp[0], e.Ss, e.Se, 0, Al});
                w.list = b->nzsets[size_t(b->nzset_of[size_t(w.image) * CSH_MAX_COMPS + size_t(e.comp[0])])].list[Al];   // made by ST_1 (Al 0..2) or ST_1B (luma Al 3)
            }
            for (uint32_t pi = sg.plan0; pi < sg.plan0 + sg.nplans; pi++) {
                TokPlan &P = b->plans[pi];
                const csh_batch::SearchImg &si = b->simg[size_t(b->plan_image[pi])];
                for (uint32_t k = 0; k < P.nslot; k++) P.s[k].Al = uint8_t(b->plan_comp[pi] == 0 ? si.Al_luma : si.Al_chroma);
            }
            if (sg.nwork) CSH_CHECK(hipMemcpyAsync(b->d_swork.p + sg.work0, b->swork.data() + sg.work0, size_t(sg.nwork) * sizeof(ScanWork), hipMemcpyHostToDevice, st));
            launch_rebind_slots(st, b->d_swork.p + sg.work0, sg.nwork, b->d_nzlists.p, b->d_slots.p);   // the slots name their list themselves (SlotRec::nzlist)
            if (sg.nplans) CSH_CHECK(hipMemcpyAsync(b->d_plans.p + sg.plan0, b->plans.data() + sg.plan0, size_t(sg.nplans) * sizeof(TokPlan), hipMemcpyHostToDevice, st));
        }
    }
    return 0;
}
// every file's list of scans: DC, luma bands, luma refinements down to the Al both share, chroma bands, chroma refinements down to it, then the
// shared refinements, luma first
static int search_lists(csh_batch *b) {
    hipStream_t st = b->stream;
    for (int i = 0; i < b->nimg; i++) {
        const csh_batch::SearchImg &si = b->simg[size_t(i)];
        const ImgDesc &im = b->imgs[size_t(i)];
        uint32_t *list = b->img_list.data() + size_t(i) * CSH_LIST_MAX;
        uint32_t m = 0;
        auto put = [&](int cand) { list[m++] = uint32_t(search_work(si, cand)); };
        const int min_Al = im.ncomp == 3 ? std::min(si.Al_luma, si.Al_chroma) : si.Al_luma;
        put(0);
        if (si.split_luma == 0) put(kLumaSplit0); else { put(kLumaSplit0 + 2 * si.split_luma - 1); put(kLumaSplit0 + 2 * si.split_luma); }
        for (int Al = si.Al_luma - 1; Al >= min_Al; Al--) put(3 + 3 * Al);
        if (im.ncomp == 3) {
            if (si.split_chroma == 0) { put(kChromaSplit0); put(kChromaSplit0 + 1); }
            else for (int k = 2; k <= 5; k++) put(kChromaSplit0 + 4 * (si.split_chroma - 1) + k);
            for (int Al = si.Al_chroma - 1; Al >= min_Al; Al--) { put(kChromaBase + 6 * Al + 4); put(kChromaBase + 6 * Al + 5); }
        }
        for (int Al = min_Al - 1; Al >= 0; Al--) {
            put(3 + 3 * Al);
            if (im.ncomp == 3) { put(kChromaBase + 6 * Al + 4); put(kChromaBase + 6 * Al + 5); }
        }
        b->img_nlist[size_t(i)] = m;
    }
    CSH_CHECK(hipMemcpyAsync(b->d_img_list.p, b->img_list.data(), b->img_list.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    CSH_CHECK(hipMemcpyAsync(b->d_img_nlist.p, b->img_nlist.data(), b->img_nlist.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    return 0;
}

static int run_once(csh_batch *b, csh_timing *t, bool requant_only) {
    hipStream_t st = b->stream;
    bool eobrun_cleared = false;   // the decode phase's store-less passes have cleared the encoder's EOBRUN array on the side (k_dec_dense clear_share)
    const int nimg = b->nimg;
    uint64_t raw_chunks = (b->raw_bytes_cap + 63) / 64;
    if (b->d_raw.n != raw_chunks * 16) {
        if (b->d_raw.alloc(raw_chunks * 16) || b->d_chunk_ff.alloc(raw_chunks + 1) || b->d_out.alloc(b->out_cap + 64))
            return -1;
        const uint64_t longest = std::max({uint64_t(size_t(b->nslots)), uint64_t(b->swork.size()), uint64_t(b->dc_total), uint64_t(b->total_sub), uint64_t(b->bits_pool.size() / 64), uint64_t(nimg)});
        size_t tmp = exclusive_scan_tmp_bytes(longest + 1);   // the longest input any exclusive scan of a run gets
        if (b->d_scan_tmp.alloc(tmp)) return -1;
    }
    if (b->d_tokens.n < b->tok_cap && b->d_tokens.alloc(b->tok_cap)) return -1;
    if (b->d_nz_pool.n < b->nz_cap && b->d_nz_pool.alloc(b->nz_cap)) return -1;
    hipEvent_t ev[CSH_NKERNELS + 1];
    for (auto &e : ev) CSH_CHECK(hipEventCreate(&e));
    int slot = 0;
    // one event after every kernel, on the batch's own stream: kernel_ms[i] = ev[i+1] - ev[i]
#define MARK() CSH_CHECK(hipEventRecord(ev[++slot], st))
    CSH_CHECK(hipEventRecord(ev[0], st));
    // a re-run at another quality (size targeting): from the retained DCT -- unless the batch derings: the overshoot mozjpeg's deringing allows
    // depends on the DC quantiser (jcdctmgr.c preprocess_deringing), so the forward DCT's input changes with the table and the re-run starts
    // at the pixel phase, from the decoded coefficients that are still in the pool
    const bool from_pixels = requant_only && b->dering;
    if (requant_only && !from_pixels) {
        if (b->d_status.zero(st) || b->d_overflow.zero(st)) return -1;
        launch_requant(st, b->d_imgs.p, b->d_pwork.p, int(b->pwork.size()), b->max_tiles, b->d_quants.p, b->d_dct_raw.p, b->ntiles_in, b->d_coef.p);
        launch_fix_dummy(st, b->d_imgs.p, nimg, b->max_dummy, b->d_coef.p);
        slot = 14;   // kernel_ms slots of the decode + pixel phases: only the first carries time (k_requant + k_fix_dummy)
        for (int s = 1; s <= slot; s++) CSH_CHECK(hipEventRecord(ev[s], st));
    } else {
    if (from_pixels) {
        if (b->d_status.zero(st) || b->d_overflow.zero(st)) return -1;
        slot = 9;   // the decode phase's kernel_ms slots stay empty
        for (int s = 1; s <= slot; s++) CSH_CHECK(hipEventRecord(ev[s], st));
    } else {
    // ---- phase 0: entropy decode (tiles must start at zero: the decoder only writes non-zero coefficients).  Where the speculation pass runs, its workgroups
    // clear the tiles on the side (k_dec_dense<0>); a memset in front of the phase otherwise (only progressive / irregular scans listed)
    const bool zero_in_spec = !b->pscans.empty() && b->max_sub != 0;
    if (!zero_in_spec) CSH_CHECK(hipMemsetAsync(b->d_coef.p, 0, size_t(b->ntiles_in) * CSH_TILE_I16 * sizeof(int16_t), st));
    if (b->d_status.zero(st) || b->d_overflow.zero(st)) return -1;
    CSH_CHECK(hipMemcpyAsync(b->d_need_seq.p, b->d_need_seq_init.p, size_t(nimg) * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    MARK();
    {   // parallel self-synchronising decode of sequential-mode scans
        int nps = int(b->pscans.size());
        uint32_t nchunks = uint32_t(b->bits_pool.size() / 64);
        if (nps) {
            launch_unstuff_count(st, b->d_bits.p, b->d_pscans.p, nps, nchunks, b->d_unstuff_cnt.p);
            launch_exclusive_scan(st, b->d_unstuff_cnt.p, b->d_unstuff_off.p, nchunks, b->d_scan_tmp.p, b->d_scan_tmp.n);
            launch_unstuff_copy(st, b->d_bits.p, b->d_clean.p, b->d_pscans.p, nps, nchunks, b->d_unstuff_off.p);
        }
        MARK();
        DenseArgs da;
        memset(&da, 0, sizeof da);
        da.clean = b->d_clean.p; da.pss = b->d_pscans.p; da.huffs = b->use4 ? static_cast<const void *>(b->d_phsets4.p) : static_cast<const void *>(b->d_phsets.p); da.compact = b->use4 ? 1 : 0; da.state = b->d_pstate.p; da.nblk = b->d_nblk.p;
        da.list_out = b->d_relax_list[0].p; da.cnt_out = b->d_relax_cnt.p; da.blk_off = b->d_blk_off.p; da.imgs = b->d_imgs.p;
        da.coef = b->d_coef.p; da.dcdiff = b->d_dcdiff.p; da.need_seq = b->d_need_seq.p; da.cut_block = b->d_cut_block.p;
        CSH_CHECK(hipMemsetAsync(b->d_cut_block.p, 0xFF, b->d_cut_block.n * sizeof(uint32_t), st));
        const uint64_t zero_all = uint64_t(b->ntiles_in) * CSH_TILE_I16 * sizeof(int16_t), zero_half = (zero_all / 2) & ~uint64_t(15);
        const uint64_t eob_all = (uint64_t(b->d_eobrun.n) * sizeof(uint16_t)) & ~uint64_t(15), eob_half = (eob_all / 2) & ~uint64_t(15);   // (the last < 16 bytes: a memset below)
        if (zero_in_spec) { da.zero_ptr = reinterpret_cast<uint8_t *>(b->d_coef.p); da.zero_bytes = zero_half; da.zero2_ptr = reinterpret_cast<uint8_t *>(b->d_eobrun.p); da.zero2_bytes = eob_half; }
        launch_dec_dense(st, 0, nps, b->max_sub, da);
        da.zero_bytes = 0; da.zero2_bytes = 0;
        MARK();
        if (nps) CSH_CHECK(hipMemsetAsync(b->d_relax_cnt.p, 0, b->d_relax_cnt.n * sizeof(uint32_t), st));
        if (nps && b->d_claim.zero(st)) return -1;
        if (zero_in_spec) {   // (the other halves: k_dec_dense<1>)
            da.zero_ptr = reinterpret_cast<uint8_t *>(b->d_coef.p) + zero_half; da.zero_bytes = zero_all - zero_half;
            da.zero2_ptr = reinterpret_cast<uint8_t *>(b->d_eobrun.p) + eob_half; da.zero2_bytes = eob_all - eob_half;
            if (uint64_t(b->d_eobrun.n) * sizeof(uint16_t) > eob_all) CSH_CHECK(hipMemsetAsync(reinterpret_cast<uint8_t *>(b->d_eobrun.p) + eob_all, 0, uint64_t(b->d_eobrun.n) * sizeof(uint16_t) - eob_all, st));
            eobrun_cleared = b->d_eobrun.n != 0;
        }
        launch_dec_dense(st, 1, nps, b->max_sub, da);
        da.zero_bytes = 0; da.zero2_bytes = 0;
        MARK();
        // list rounds until the list is empty.  How many that takes depends on the data: stock tables at ordinary quality settle
        // in ~8 (the list shrinks by 60 % a round), 50 bytes per block in ~30, 85 bytes per block in more than a hundred (a
        // wrong state then survives most of the cuts it crosses) -- so the host looks at the list length after 12 rounds and
        // then after every 8 (a 4-byte read-back; an empty round is a ~6 us launch), up to kMaxRounds.  What is still listed
        // after that goes through the label chain below or to the sequential kernel.
        const int kMaxRounds = int(b->d_relax_cnt.n) - 2;
        int R = 0;
        for (int group = 12; nps && R < kMaxRounds; group = 8) {
            for (int g = 0; g < group && R < kMaxRounds; g++, R++)
                launch_dec_relax_list(st, b->d_clean.p, b->d_pscans.p, b->total_sub, da.huffs, da.compact, b->d_pstate.p, b->d_nblk.p, b->d_relax_list[R & 1].p,
                                      b->d_relax_cnt.p + R, b->d_relax_list[(R & 1) ^ 1].p, b->d_relax_cnt.p + R + 1, b->d_pstate.n, b->d_claim.p, uint32_t(R + 1));
            uint32_t left = 0;
            CSH_CHECK(hipMemcpyAsync(&left, b->d_relax_cnt.p + R, sizeof left, hipMemcpyDeviceToHost, st));
            CSH_CHECK(hipStreamSynchronize(st));
            if (!left) break;
        }
        if (nps) {   // scans that are still listed: settle their block-in-MCU labels exactly (k_dec_chain), or hand the image to k_decode_seq
            if (b->d_scan_pending.zero(st)) return -1;
            launch_dec_mark_pending(st, b->d_pscans.p, b->total_sub, b->d_relax_list[R & 1].p, b->d_relax_cnt.p + R, b->d_scan_pending.p);
            da.hyp = b->d_hyp.p; da.scan_pending = b->d_scan_pending.p;
            launch_dec_dense(st, 3, nps, b->max_sub, da);
            launch_dec_chain(st, b->d_pscans.p, nps, b->d_pstate.p, b->d_nblk.p, b->d_hyp.p, b->d_scan_pending.p, b->d_need_seq.p);
        }
        MARK();
        if (nps) launch_exclusive_scan(st, b->d_nblk.p, b->d_blk_off.p, b->total_sub, b->d_scan_tmp.p, b->d_scan_tmp.n);
        launch_dec_dense(st, 2, nps, b->max_sub, da);
        MARK();
        if (nps) launch_exclusive_scan(st, reinterpret_cast<uint32_t *>(b->d_dcdiff.p), b->d_dc_off.p, b->dc_total, b->d_scan_tmp.p, b->d_scan_tmp.n);
        launch_dc_scatter(st, b->d_pscans.p, nps, b->max_par_blocks, b->d_imgs.p, b->d_dc_off.p, b->d_coef.p, b->d_need_seq.p, b->d_cut_block.p);
        launch_dc_refine(st, b->d_clean.p, b->d_pscans.p, nps, b->max_par_blocks, b->d_imgs.p, b->d_coef.p, b->d_need_seq.p);
        MARK();
    }
    launch_refine_chains(st, b->d_clean.p, b->d_pscans.p, b->d_phsets.p, b->d_dscans.p, b->d_chains.p, b->d_chain_scans.p, int(b->chains.size()), b->d_refine_units.p,
                         int(b->refine_units.size()), b->refine_max_blocks, b->d_imgs.p, b->d_coef.p, b->d_need_seq.p, b->d_refine_hist.p, b->d_refine_pos.p, b->d_refine_prog.p);
    MARK();
    launch_decode_prog(st, b->d_clean.p, b->d_pscans.p, b->d_phsets.p, b->d_dscans.p, b->d_chains.p, b->d_chain_scans.p, int(b->chains.size()), b->d_imgs.p,
                       b->d_coef.p, b->d_need_seq.p);
    launch_decode_seq(st, b->d_bits.p, b->d_imgs.p, b->d_dscans.p, b->d_hsets.p, b->d_coef.p, nimg, b->d_need_seq.p);
    MARK();
    }  // !from_pixels
    // ---- phase 1: pixel-domain transcode
    int nw = b->lossless ? 0 : int(b->pwork.size());
    launch_idct_plane(st, b->d_imgs.p, b->d_pwork.p, nw, b->max_tiles, b->d_quants.p, b->d_coef.p, b->d_planes.p);
    MARK();
    launch_resize(st, b->d_imgs.p, b->d_rwork.p, int(b->rwork.size()), b->d_rtaps.p, b->d_rweights.p, b->d_planes.p, b->d_rgb.p, b->d_rtmp.p,
                  b->max_src_px, b->max_tmp, b->max_dst, b->max_row_in, b->max_out_w, b->max_nh, !(b->webp || b->rgb_out));
    MARK();
    if (b->webp) return run_webp(b, t, ev, slot);
    if (b->rgb_out) return run_rgb_only(b, t, ev, slot);
    int16_t *rawp = ((b->retain_dct || b->trellis) && !b->lossless) ? b->d_dct_raw.p : nullptr;   // the trellis quantiser works from the unquantised DCT
    launch_xform_direct(st, b->d_imgs.p, b->d_pwork.p, nw, b->max_tiles, b->d_quants.p, b->d_coef.p, b->d_coef.p, rawp, b->ntiles_in, b->dering);
    MARK();
    launch_resample_plane(st, b->d_imgs.p, b->d_pwork.p, nw, b->max_quads, b->d_planes.p, b->d_oplanes.p);
    launch_plane_fdct(st, b->d_imgs.p, b->d_pwork.p, nw, b->max_tiles, b->d_quants.p, b->d_oplanes.p, b->d_coef.p, rawp, b->ntiles_in, b->dering);
    launch_resample_fdct_420(st, b->d_imgs.p, b->d_pwork.p, nw, b->max_tiles, b->d_quants.p, b->d_planes.p, b->d_coef.p, rawp, b->ntiles_in, b->dering);
    MARK();
    if (!b->lossless) launch_fix_dummy(st, b->d_imgs.p, nimg, b->max_dummy, b->d_coef.p);
    MARK();
    }  // !requant_only
    // ---- phase 2: tokens (+ flags + statistics), EOB runs
    EncCtx c;
    memset(&c, 0, sizeof c);
    c.imgs = b->d_imgs.p; c.script = b->d_script.p; c.work = b->d_swork.p; c.nwork = int(b->swork.size());
    c.echunks = b->d_echunks.p; c.plans = b->d_plans.p; c.nechunks = uint32_t(b->echunks.size()); c.slot_work = b->d_slot_work.p; c.slots = b->d_slots.p; c.nslots = b->nslots;
    c.coef = b->d_coef.p; c.sym_bits = b->d_symbits.p; c.eob_bits = b->d_eobbits.p; c.tail = b->d_tail.p;
    c.eobrun = b->d_eobrun.p; c.long_runs = b->d_long_runs.p; c.long_cnt = b->d_long_cnt.p; c.corr = b->d_corr.p;
    c.tokens = b->d_tokens.p; c.regions = b->d_regions.p; c.tok_cursor = b->d_tok_cursor.p; c.tok_off = b->d_tok_off.p; c.chunk_ntok = b->d_chunk_ntok.p; c.slot_hist = b->d_slot_hist.p; c.slot_raw = b->d_slot_raw.p; c.slot_eobh = b->d_slot_eobh.p;
    c.chunk_bits = b->d_chunk_bits.p; c.chunk_off = b->d_chunk_off.p; c.tables = b->d_tables.p;
    c.raw = b->d_raw.p; c.raw_words = raw_chunks * 16; c.status = b->d_status.p; c.overflow = b->d_overflow.p;
    c.nzlists = b->d_nzlists.p; c.nzsets = b->d_nzsets.p; c.nz_pool = b->d_nz_pool.p; c.nz_cursor = b->d_nz_cursor.p; c.nz_chunk_off = b->d_nz_chunk_off.p; c.nz_chunk_cnt = b->d_nz_chunk_cnt.p;
    c.debug = getenv("CSH_DEBUG") ? uint32_t(atoi(getenv("CSH_DEBUG"))) : 0u;
    launch_reset_works(st, b->d_swork.p, c.nwork);
    if (b->d_nz_cursor.zero(st) || b->d_nz_chunk_cnt.zero(st)) return -1;
    if (b->d_symbits.zero(st) || b->d_eobbits.zero(st) || (!eobrun_cleared && b->d_eobrun.zero(st)) || b->d_tables.zero(st) || b->d_tok_cursor.zero(st) || b->d_slot_eobh.zero(st) || b->d_scan_pad.zero(st)) return -1;
#ifdef CSH_EMUL
    if (b->d_raw.zero(st)) return -1;   // the emulation's packer ORs every word into the pool (no LDS window there)
#else
    if ((c.debug & 8192u) && b->d_raw.zero(st)) return -1;
#endif
    MARK();
    // ---- mozjpeg's trellis quantiser (CSH_PROFILE=mozjpeg): per component a statistics scan over the scalar-quantised coefficients
    // (tokens without tokens: histograms, flags, EOB runs -> optimal tables), then every block re-quantised from the retained DCT
    auto set_stage = [&](const csh_batch::Stage &sg) {
        c.echunks = b->d_echunks.p + sg.ech0; c.nechunks = sg.nech; c.slot0 = sg.slot0; c.nslots = sg.nslots;
        c.nzchunks = b->d_nzchunks.p + sg.nzc0; c.nnzchunks = sg.nnzc;
        c.list_slots = b->d_list_slots.p + sg.ls0; c.nlist_slots = sg.nls; c.tok_slots = b->d_tok_slots.p + sg.ts0; c.ntok_slots = sg.nts;
    };
    if (b->trellis) {
        const csh_batch::Stage &tg = b->tstage;
        set_stage(tg);
        c.stats_only = 1;
        if (b->d_long_cnt.zero(st)) return -1;
        c.nz_blk_cnt = b->t_sort ? b->d_tblk_cnt.p : nullptr;
        c.nz_blk_off = (b->t_sort && b->nz_once) ? b->d_tblk_off.p : nullptr;
        launch_nzlist(st, c);       // level 0 of the scalar-quantised coefficients (progressive output: the statistics scans are list slots)
        c.nz_blk_cnt = nullptr; c.nz_blk_off = nullptr;
        if (b->t_sort) {   // the blocks of every component in order of list length (timed with the statistics)
            TrellisCtx ts;
            memset(&ts, 0, sizeof ts);
            ts.work = b->d_twork.p; ts.nwork = int(b->twork.size()); ts.blk_cnt = b->d_tblk_cnt.p; ts.perm = b->d_tperm.p;
            launch_trellis_sort(st, ts);
        }
        launch_tokens(st, c);       // (sequential output: one-component sequential scans, histograms only)
        launch_list_stats(st, c);
        launch_ac_runs(st, c);
        launch_gen_tables(st, b->d_tables.p + tg.table0, int(tg.ntables));
        c.stats_only = 0;
        if (b->d_nz_cursor.zero(st)) return -1;   // the lists are made again from what the trellis leaves
        MARK();
        TrellisCtx tc;
        memset(&tc, 0, sizeof tc);
        tc.imgs = b->d_imgs.p; tc.quant = b->d_quants.p; tc.work = b->d_twork.p; tc.nwork = int(b->twork.size()); tc.chunks = b->d_tchunks.p; tc.nchunks = uint32_t(b->tchunks.size());
        tc.tables = b->d_tables.p; tc.raw = b->d_dct_raw.p; tc.raw_tile0 = b->ntiles_in; tc.coef = b->d_coef.p; tc.dcrec = b->d_tlambda.p; tc.dcbt = b->d_tdcbt.p;
        tc.spill = b->d_tspill.p; tc.max_rows = b->t_max_rows;
        tc.rows = b->d_trows.p; tc.nrows = uint32_t(b->trows.size());
        if (b->t_sort) { tc.blk_cnt = b->d_tblk_cnt.p; tc.perm = b->d_tperm.p; }
        if (b->t_sort && b->nz_once) {
            tc.nz_pool = b->d_nz_pool.p; tc.nzlists = b->d_nzlists.p; tc.nzsets = b->d_nzsets.p; tc.nz_chunk_off = b->d_nz_chunk_off.p; tc.nz_chunk_cnt = b->d_nz_chunk_cnt.p;
            tc.blk_off = b->d_tblk_off.p;
        }
        tc.debug = getenv("CSH_TR_DEBUG") ? uint32_t(atoi(getenv("CSH_TR_DEBUG"))) : 0u;
        launch_trellis_ac(st, tc);
        MARK();
        launch_trellis_dc(st, tc);
        launch_fix_dummy(st, b->d_imgs.p, nimg, b->max_dummy, b->d_coef.p);   // the dummy blocks copy DC values the trellis has just changed
        MARK();
    } else { MARK(); MARK(); MARK(); }
    AsmCtx a;
    memset(&a, 0, sizeof a);
    a.imgs = b->d_imgs.p; a.script = b->d_script.p; a.work = b->d_swork.p; a.nimg = nimg;
    a.nwork = b->trellis ? int(b->tstage.work0) : c.nwork;   // the trellis stage's statistics scans (the last work items) put nothing into a file
    a.tables = b->d_tables.p; a.chunk_off = b->d_chunk_off.p; a.scan_pad_bytes = b->d_scan_pad.p; a.scan_raw_off = b->d_scan_raw_off.p;
    a.raw = b->d_raw.p; a.raw_chunks = raw_chunks; a.chunk_ff = b->d_chunk_ff.p;
    a.hdr_pool = b->d_hdr.p; a.hdr_off = b->d_hdr_off.p; a.img_size = b->d_img_size.p; a.img_size_pad = b->d_img_size_pad.p;
    a.img_off = b->d_img_off.p; a.out = b->d_out.p; a.out_cap = b->out_cap; a.status = b->d_status.p; a.overflow = b->d_overflow.p;
    a.img_list = b->d_img_list.p; a.img_nlist = b->d_img_nlist.p; a.scan_cost = b->d_scan_cost.p;
    // one stage = tokens -> runs -> tables -> chunk sizes -> offsets -> pack -> stuffing counts, over a contiguous range of work items
    // (without the scan search: one stage, everything).  mark: timing slots are recorded for stage 1 only, stage 2 gets one slot.
    auto run_stage = [&](const csh_batch::Stage &sg, bool mark, bool gate) -> int {
#define SMARK() do { if (mark) MARK(); } while (0)
        set_stage(sg);
        c.work_active = gate ? b->d_work_active.p : nullptr;
        a.work0 = int(sg.work0); a.nwork_run = int(sg.nwork);
        if (b->d_long_cnt.zero(st)) return -1;
        launch_nzlist(st, c);       // the lists this stage's first-pass scans are coded from and no earlier stage made
        SMARK();
        launch_tokens(st, c);       // DC, sequential-mode and refinement scans
        SMARK();
        launch_list_stats(st, c);   // AC first-pass scans
        SMARK();
        launch_ac_runs(st, c);
        SMARK();
        launch_gen_tables(st, b->d_tables.p + sg.table0, int(sg.ntables));
        SMARK();
        launch_chunk_sizes(st, c);
        SMARK();
        launch_exclusive_scan(st, b->d_chunk_bits.p + sg.slot0, b->d_chunk_off.p + sg.slot0, sg.nslots, b->d_scan_tmp.p, b->d_scan_tmp.n);
        SMARK();
        launch_scan_sizes(st, a);
        launch_exclusive_scan(st, b->d_scan_pad.p, b->d_scan_raw_off.p, uint64_t(a.nwork), b->d_scan_tmp.p, b->d_scan_tmp.n);   // all work items: those of a later stage still count zero
        launch_scan_place(st, a);
        launch_zero_edges(st, c);
        SMARK();
        launch_pack(st, c);
        SMARK();
        launch_list_pack(st, c);
        SMARK();
        launch_ff_count(st, a);
        SMARK();
#undef SMARK
        return 0;
    };
    if (run_stage(b->stage[0], true, false)) return -1;
    if (b->search) {
        b->n_gated_runs = 0;
        // a conditional stage: coded only if some image's search asks for it, and then only for those images (work_active)
        auto gated = [&](int sid, auto want) -> int {
            if (!search_gate(b, sid, want)) return 0;
            b->n_gated_runs++;
            if (b->d_work_active.upload(b->work_active, st)) return -1;
            if (run_stage(b->stage[sid], false, true) || search_costs(b, a, sid)) return -1;
            return search_decide(b, sid);
        };
        if (search_costs(b, a, csh_batch::ST_1) || search_decide(b, csh_batch::ST_1)) return -1;      // Al of luma (unless Al 3 is still to be tried) and of chroma
        if (gated(csh_batch::ST_1B, [&](int i) { return b->simg[size_t(i)].luma_on ? 1 : 0; })) return -1;
        if (run_stage(b->stage[csh_batch::ST_2], false, false)) return -1;
        if (search_costs(b, a, csh_batch::ST_2) || search_decide(b, csh_batch::ST_2)) return -1;      // the splits up to the third
        for (int sid : {int(csh_batch::ST_2B), int(csh_batch::ST_2C)})
            if (gated(sid, [&](int i) { return (b->simg[size_t(i)].luma_on ? 1 : 0) | (b->simg[size_t(i)].chroma_on ? 2 : 0); })) return -1;
        if (search_lists(b)) return -1;
    }
    MARK();
    launch_layout(st, a);
    MARK();
    launch_exclusive_scan(st, b->d_img_size_pad.p, b->d_img_off.p, uint64_t(nimg), b->d_scan_tmp.p, b->d_scan_tmp.n);
    MARK();
    launch_emit(st, a);
    MARK();
#undef MARK
    CSH_CHECK(hipStreamSynchronize(st));
    CSH_CHECK(hipGetLastError());
    if (t) {
        for (int i = 0; i < CSH_NPHASES; i++) t->phase_ms[i] = 0;
        for (int i = 0; i < CSH_NKERNELS; i++) t->kernel_ms[i] = 0;
        for (int i = 0; i < slot; i++) {
            CSH_CHECK(hipEventElapsedTime(&t->kernel_ms[i], ev[i], ev[i + 1]));
            t->phase_ms[kKernelPhase[i]] += t->kernel_ms[i];
        }
        CSH_CHECK(hipEventElapsedTime(&t->total_ms, ev[0], ev[slot]));
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    return 0;
}

static int batch_run(csh_batch *b, csh_timing *t, bool requant_only);
extern "C" int csh_batch_run(csh_batch *b, csh_timing *t) { return batch_run(b, t, false); }

// size targeting (caesium::compress_to_size_in_memory, compressor.rs:295,298): keep the unquantised DCT of the first run ...
extern "C" int csh_batch_retain_dct(csh_batch *b, int on) {
    if (b->lossless) { csh_set_error("retain_dct: a coefficient transcode has no quality to re-target"); return -1; }
    b->retain_dct = on != 0;
    if (b->retain_dct && b->nimg && b->d_dct_raw.n == 0 && b->d_dct_raw.alloc(size_t(b->ntiles_out) * CSH_TILE_I16)) return -1;
    return 0;
}
// ... give some images another quality (quality[i] == 0: unchanged; indexed like the inputs) ...
extern "C" int csh_batch_set_quality(csh_batch *b, const uint32_t *quality) {
    if (b->lossless) { csh_set_error("set_quality on a lossless batch"); return -1; }
    b->hdr_pool.clear(); b->hdr_off.clear();
    for (size_t n = 0; n < b->items.size(); n++) {
        Item &it = b->items[n];
        if (it.image < 0) continue;
        ImgDesc &im = b->imgs[it.image];
        if (quality[n]) {
            int q = int(quality[n]) < 1 ? 1 : (quality[n] > 100 ? 100 : int(quality[n]));
            for (int c = 0; c < im.ncomp; c++) im.qt_out[c] = b->q_base + q;
        }
        JpegInfo hdr = it.out;
        int qidx = im.qt_out[0];
        uint16_t nat[64];
        for (int k = 0; k < 64; k++) nat[kZigZag[k]] = b->quants[qidx].q[k];
        memcpy(hdr.qt[0], nat, 128); memcpy(hdr.qt[1], nat, 128);
        std::vector<uint8_t> fh = build_frame_header(hdr, b->progressive, it.meta_out.empty() ? nullptr : &it.meta_out);
        b->hdr_off.push_back(uint32_t(b->hdr_pool.size()));
        b->hdr_pool.insert(b->hdr_pool.end(), fh.begin(), fh.end());
    }
    b->hdr_off.push_back(uint32_t(b->hdr_pool.size()));
    if (!b->nimg) return 0;
    if (hipSetDevice(b->device) != hipSuccess) return -1;
    if (b->d_imgs.upload(b->imgs, b->stream) || b->d_hdr.upload(b->hdr_pool, b->stream) || b->d_hdr_off.upload(b->hdr_off, b->stream)) return -1;
    CSH_CHECK(hipStreamSynchronize(b->stream));
    return 0;
}
// ... and re-run only re-quantisation + entropy coding + assembly.
extern "C" int csh_batch_rerun_encode(csh_batch *b, csh_timing *t) {
    if (!b->have_dct) { csh_set_error("rerun_encode needs a completed csh_batch_run after csh_batch_retain_dct(1)"); return -1; }
    return batch_run(b, t, true);
}

static int batch_run(csh_batch *b, csh_timing *t, bool requant_only) {
    if (t) memset(t, 0, sizeof *t);
    if (!b->nimg) { b->ran = true; return 0; }
    if (hipSetDevice(b->device) != hipSuccess) { csh_set_error("hipSetDevice failed"); return CS_ERR_NO_DEVICE; }
    for (int attempt = 0; attempt < 4; attempt++) {
        if (run_once(b, t, requant_only)) return CS_ERR_NO_DEVICE;
        uint32_t ovf[4] = {0, 0, 0, 0};
        if (csh_copy_wait(ovf, b->d_overflow.p, sizeof ovf, hipMemcpyDeviceToHost, b->stream) != hipSuccess) { csh_set_error("D2H failed"); return CS_ERR_NO_DEVICE; }
        b->h_status.resize(b->nimg);
        if (csh_copy_wait(b->h_status.data(), b->d_status.p, b->nimg * sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream) != hipSuccess) return CS_ERR_NO_DEVICE;
        bool pool = ovf[0] != 0;
        if (ovf[1]) {   // token pool (k_tokens)
            pool = true; b->tok_scale *= 4;
            layout_token_pool(b);
            if (b->d_regions.upload(b->regions, b->stream) || b->d_nzlists.upload(b->nzlists, b->stream) || hipStreamSynchronize(b->stream) != hipSuccess) return CS_ERR_NO_DEVICE;
        }
        for (uint32_t s : b->h_status) if (s == CS_ERR_POOL_OVERFLOW) pool = true;
        if (!pool) break;
        if (attempt == 3) { csh_set_error("device pools overflowed after 3 retries"); return CS_ERR_POOL_OVERFLOW; }
        b->raw_bytes_cap *= 4; b->out_cap = b->raw_bytes_cap;  // rare: output larger than 2x the input
        b->webp_mb_bytes *= 4;
    }
    b->h_img_size.resize(b->nimg);
    b->h_img_off.resize(b->nimg + 1);
    if (csh_copy_wait(b->h_img_size.data(), b->d_img_size.p, b->nimg * sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream) != hipSuccess ||
        csh_copy_wait(b->h_img_off.data(), b->d_img_off.p, (b->nimg + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, b->stream) != hipSuccess) {
        csh_set_error("D2H of sizes failed"); return CS_ERR_NO_DEVICE;
    }
    if (t) {
        std::vector<uint32_t> ns(b->nimg);
        if (csh_copy_wait(ns.data(), b->d_need_seq.p, b->nimg * sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream) != hipSuccess) return CS_ERR_NO_DEVICE;
        if (getenv("CSH_TRACE") && b->d_relax_cnt.n) {   // sub-sequences re-listed after each relaxation round
            std::vector<uint32_t> rc(b->d_relax_cnt.n);
            if (csh_copy_wait(rc.data(), b->d_relax_cnt.p, rc.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream) == hipSuccess) {
                fprintf(stderr, "[csh] relax list sizes (of %u sub-sequences):", b->total_sub);
                for (size_t i = 0; i < rc.size() && (i < 16 || rc[i]); i++) fprintf(stderr, " %u", rc[i]);
                fprintf(stderr, "\n");
            }
        }
        for (const ProgChain &pc : b->chains) if (pc.refine && ns[size_t(pc.image)] == 4) t->n_refine_chains++;
        for (uint32_t v : ns) { if (v == 4) { t->n_prog_decoded++; continue; } if (v) t->n_seq_decoded++; if (v == 2 || v == 3) t->n_par_fallback++; if (v == 3) t->n_par_short++; }
        t->n_images = uint32_t(b->nimg);
        t->n_search_extra = b->search ? b->n_gated_runs : 0u;
        for (const Item &it : b->items) if (it.image < 0) t->n_failed++;
        for (int i = 0; i < b->nimg; i++) { t->out_bytes += b->h_img_size[i]; t->pixels += uint64_t(b->imgs[i].width) * b->imgs[i].height; }
        t->in_bytes = b->bits_pool.size();
        uint64_t in_tiles = 0;
        for (const ImgDesc &im : b->imgs) for (int c = 0; c < im.ncomp; c++) in_tiles += im.in[c].ntiles;
        t->coef_bytes = in_tiles * CSH_TILE_I16 * 2;
    }
    b->ran = true;
    if (!requant_only) b->have_dct = (b->retain_dct || b->trellis) && !b->lossless;
    return 0;
}

static void set_result(CCSResult *r, int code, const std::string &msg) {
    r->success = code == 0;
    r->code = uint32_t(code);
    r->error_message = nullptr;
    if (code) { char *m = (char *)malloc(msg.size() + 1); memcpy(m, msg.c_str(), msg.size() + 1); r->error_message = m; }
}

// the decoded (and resized) image of a csh_batch_create_pixels batch, still in device memory: interleaved 8-bit samples, 1 or 3 per pixel.
// Returns the file's CCSResult code (0: the pointers are set; they live as long as the batch)
extern "C" int csh_batch_pixels(csh_batch *b, size_t image, const uint8_t **device_pixels, uint32_t *width, uint32_t *height, uint32_t *channels, const char **message) {
    *device_pixels = nullptr; *width = *height = *channels = 0;
    if (message) *message = "";
    if (!b || !b->ran || !b->rgb_out || image >= b->items.size()) { csh_set_error("csh_batch_pixels: not a pixel batch that has run"); if (message) *message = "not a pixel batch that has run"; return CS_ERR_NO_DEVICE; }
    const Item &it = b->items[image];
    if (it.code) { if (message) *message = it.msg.c_str(); return it.code; }
    if (b->h_status[it.image]) { if (message) *message = "device reported a malformed stream"; return int(b->h_status[it.image]); }
    if (size_t(it.image) < b->rwork.size() && b->rwork[size_t(it.image)].image == it.image) {   // every image of a pixel batch has its resize work item, in image order
        const ResizeWork &rw = b->rwork[size_t(it.image)];
        *device_pixels = b->d_rgb.p + rw.rgb_dst_off; *width = uint32_t(rw.nw); *height = uint32_t(rw.nh); *channels = uint32_t(b->imgs[it.image].ncomp);
        return 0;
    }
    if (message) *message = "image has no pixel output";
    return CS_ERR_NO_DEVICE;
}

extern "C" int csh_batch_fetch(csh_batch *b, CByteArray *outputs, CCSResult *results) {
    if (!b->ran) { csh_set_error("csh_batch_fetch before csh_batch_run"); return -1; }
    if (b->rgb_out) { csh_set_error("csh_batch_fetch: a pixel batch has no files (csh_batch_pixels)"); return -1; }
    struct PinnedOut { uint8_t *p = nullptr; size_t cap = 0; ~PinnedOut() { if (p) pinned_cache().put(p, cap); } uint8_t *data() const { return p; } } host;
    if (b->nimg && b->h_img_off[b->nimg]) {
        host.p = static_cast<uint8_t *>(pinned_cache().get(b->h_img_off[b->nimg], host.cap));
        if (!host.p) { csh_set_error("out of pinned host memory"); return -1; }
        if (csh_copy_wait(host.p, b->d_out.p, b->h_img_off[b->nimg], hipMemcpyDeviceToHost, b->stream) != hipSuccess) { csh_set_error("D2H of output failed"); return -1; }
    }
    std::atomic<int> failed{0};
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (size_t n; (n = next++) < b->items.size();) {
            Item &it = b->items[n];
            outputs[n].data = nullptr; outputs[n].length = 0;
            int code = it.code;
            std::string msg = it.msg;
            if (!code && it.image >= 0 && b->h_status[it.image]) { code = int(b->h_status[it.image]); msg = "device reported a malformed stream"; }
            if (!code) {
                size_t len = b->h_img_size[it.image];
                outputs[n].data = (uint8_t *)malloc(len ? len : 1);
                memcpy(outputs[n].data, host.data() + b->h_img_off[it.image], len);
                outputs[n].length = len;
            } else failed++;
            if (results) set_result(&results[n], code, msg);
        }
    };
    size_t nthreads = std::min<size_t>(std::min<size_t>(8, std::max(1u, std::thread::hardware_concurrency())), b->items.size() / 64 + 1);
    std::vector<std::thread> pool;
    for (size_t t = 1; t < nthreads; t++) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    return failed.load();
}

extern "C" int csh_batch_geometry(csh_batch *b, size_t image, int comp, int which, int *bw, int *bh, int *real_bw, int *real_bh) {
    if (image >= b->items.size() || b->items[image].image < 0) { csh_set_error("image not on the device"); return -1; }
    const ImgDesc &im = b->imgs[b->items[image].image];
    if (comp < 0 || comp >= im.ncomp) { csh_set_error("bad component"); return -1; }
    const CompGeom &g = which ? im.out[comp] : im.in[comp];
    *bw = g.bw; *bh = g.bh; *real_bw = g.real_bw; *real_bh = g.real_bh;
    return 0;
}

extern "C" int csh_batch_read_coefs(csh_batch *b, size_t image, int comp, int which, int16_t *dst) {
    int bw, bh, rbw, rbh;
    if (csh_batch_geometry(b, image, comp, which, &bw, &bh, &rbw, &rbh)) return -1;
    const ImgDesc &im = b->imgs[b->items[image].image];
    const CompGeom &g = which ? im.out[comp] : im.in[comp];
    std::vector<int16_t> tiles(size_t(g.ntiles) * CSH_TILE_I16);
    if (csh_copy_wait(tiles.data(), b->d_coef.p + size_t(g.tile_base) * CSH_TILE_I16, tiles.size() * 2, hipMemcpyDeviceToHost, b->stream) != hipSuccess) { csh_set_error("D2H failed"); return -1; }
    for (int blk = 0; blk < bw * bh; blk++)
        for (int k = 0; k < 64; k++) dst[size_t(blk) * 64 + k] = tiles[size_t(blk >> 6) * CSH_TILE_I16 + (blk & 63) * CSH_BLK_STRIDE + coef_off(k)];
    return 0;
}
