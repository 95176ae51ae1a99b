// k_assemble.hip -- phase 6: turn the packed scan bit strings into finished JPEG files in HBM:
// 0xFF byte stuffing (T.81 F.1.2.3), DHT/SOS headers per scan (mozjpeg's merged-DHT marker style, as
// /root/reference/samples/j0.JPG shows), EOI.  Replaces mozjpeg's jcmarker.c + the byte-level part of
// jchuff/jcphuff for libcaesium's JPEG path (reference call site /root/reference/src/compressor.rs:305).
// Every offset is computed on the device (exclusive scans), so the whole batch needs no host round trip
// between entropy coding and the final D2H copy.
#include "kernels.h"
#include "wave.h"

namespace csh {

// ------------------------------------------------------------------------------------------------
// out[i] = sum of in[0 .. i) for i = 0 .. n (n + 1 outputs of 64 bits, n inputs of 32): the one scan primitive of the pipeline (chunk
// offsets, scan offsets, block ordinals, DC prefixes: ~2 k image sizes up to ~100 M DC differences).  Three launches: sums of
// 16384-element stretches (a wave sums a quarter, 64 lanes x 64 coalesced steps), an exclusive scan of those sums by one workgroup, and the
// stretches again with their offsets -- a wave scan (DPP) per step, the running total carried in a scalar.
#define CSH_SCAN_WAVE 4096                       // elements per wave
#define CSH_SCAN_STRETCH (4 * CSH_SCAN_WAVE)     // per workgroup
__device__ __forceinline__ static uint64_t scan_wave_sum(const uint32_t *in, uint64_t n, uint64_t w0) {
    LV<uint64_t> acc;
    LFOR(l) {
        uint64_t a = 0;
        for (uint32_t k = 0; k < CSH_SCAN_WAVE / 64; k++) { const uint64_t i = w0 + uint64_t(k) * 64 + uint64_t(l); if (i < n) a += in[i]; }
        acc[l] = a;
    }
    return csp::lsum(acc);
}
__global__ void __launch_bounds__(256) k_scan_sums(const uint32_t *in, uint64_t n, uint64_t *sums) {
    CSH_SHARED uint64_t s_part[4];
    const int wv = int(threadIdx.x) / CSP_WAVE_THREADS;
    CSH_PHASE_LOOP(2) {
        if (phase == 0) {
            const uint64_t t = scan_wave_sum(in, n, uint64_t(blockIdx.x) * CSH_SCAN_STRETCH + uint64_t(wv) * CSH_SCAN_WAVE);
            LFOR(l) if (l == 0) s_part[wv] = t;
            continue;
        }
        if (wv == 0) LFOR(l) if (l == 0) sums[blockIdx.x] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    }
}
__global__ void __launch_bounds__(256) k_scan_spine(uint64_t *sums, uint32_t nb) {   // in place: sums[b] = sum of the stretches in front of b; sums[nb] = everything
    CSH_SHARED uint64_t s_part[257];
    const uint32_t per = (nb + 255) / 256, lo = min(nb, threadIdx.x * per), hi = min(nb, lo + per);
    CSH_PHASE_LOOP(3) {
        if (phase == 0) { uint64_t acc = 0; for (uint32_t b = lo; b < hi; b++) acc += sums[b]; s_part[threadIdx.x] = acc; continue; }
        if (phase == 1) { if (threadIdx.x == 0) { uint64_t t = 0; for (int k = 0; k < 256; k++) { const uint64_t v = s_part[k]; s_part[k] = t; t += v; } s_part[256] = t; } continue; }
        uint64_t acc = s_part[threadIdx.x];
        for (uint32_t b = lo; b < hi; b++) { const uint64_t v = sums[b]; sums[b] = acc; acc += v; }
        if (threadIdx.x == 0) sums[nb] = s_part[256];
    }
}
__global__ void __launch_bounds__(256) k_scan_down(const uint32_t *in, uint64_t n, const uint64_t *sums, uint32_t nb, uint64_t *out) {
    CSH_SHARED uint64_t s_part[4];
    const int wv = int(threadIdx.x) / CSP_WAVE_THREADS;
    const uint64_t w0 = uint64_t(blockIdx.x) * CSH_SCAN_STRETCH + uint64_t(wv) * CSH_SCAN_WAVE;
    CSH_PHASE_LOOP(2) {
        if (phase == 0) {
            const uint64_t t = scan_wave_sum(in, n, w0);
            LFOR(l) if (l == 0) s_part[wv] = t;
            continue;
        }
        uint64_t carry = sums[blockIdx.x];
        for (int q = 0; q < wv; q++) carry += s_part[q];
        for (uint32_t k = 0; k < CSH_SCAN_WAVE / 64 && w0 + uint64_t(k) * 64 < n; k++) {
            LV<uint32_t> v;
            LFOR(l) { const uint64_t i = w0 + uint64_t(k) * 64 + uint64_t(l); v[l] = i < n ? in[i] : 0u; }
            uint32_t tot;
            const LV<uint32_t> ex = lscan(v, tot);   // (a step's 64 inputs sum below 2^32 wherever this scan is used: counts and sizes)
            LFOR(l) { const uint64_t i = w0 + uint64_t(k) * 64 + uint64_t(l); if (i < n) out[i] = carry + ex[l]; }
            carry += tot;
        }
        if (blockIdx.x == nb - 1 && wv == 3) LFOR(l) if (l == 0) out[n] = sums[nb];
    }
}
size_t exclusive_scan_tmp_bytes(uint64_t n) { return size_t((n + CSH_SCAN_STRETCH - 1) / CSH_SCAN_STRETCH + 2) * sizeof(uint64_t); }
// CONTRACT: out[i] is the true 64-bit prefix wherever every stretch of 64 consecutive inputs sums below 2^32 (sizes, counts: every caller but one).  The DC
// differences (pipeline.cpp: dcdiff reinterpreted as uint32_t, negative values ~ 2^32) break that premise on purpose: their prefix is only right modulo
// 2^32, and k_dc_scatter uses the low 32 bits of the DIFFERENCE of two prefixes only.
void launch_exclusive_scan(hipStream_t st, const uint32_t *in, uint64_t *out, uint64_t n, void *tmp, size_t tmp_bytes) {
    uint64_t *sums = static_cast<uint64_t *>(tmp);
    const uint32_t nb = uint32_t((n + CSH_SCAN_STRETCH - 1) / CSH_SCAN_STRETCH);
    if (!nb) { (void)hipMemsetAsync(out, 0, sizeof(uint64_t), st); return; }
    // (cannot happen: every caller sizes tmp with exclusive_scan_tmp_bytes for its largest n.  If it does: the offsets are all zero -- every consumer stays
    // inside its pool -- and the error string says why the run's output is wrong)
    if (size_t(nb + 2) * sizeof(uint64_t) > tmp_bytes) { csh_set_error("exclusive scan: temporary buffer too small"); (void)hipMemsetAsync(out, 0, size_t(n + 1) * sizeof(uint64_t), st); return; }
    CSH_LAUNCH_PHASED(k_scan_sums, 2, dim3(nb), dim3(4 * CSP_WAVE_THREADS), st, in, n, sums);
    CSH_LAUNCH_PHASED(k_scan_spine, 3, dim3(1), dim3(256), st, sums, nb);
    CSH_LAUNCH_PHASED(k_scan_down, 2, dim3(nb), dim3(4 * CSP_WAVE_THREADS), st, in, n, sums, nb, out);
}

// ------------------------------------------------------------------------------------------------
__global__ void k_scan_sizes(AsmCtx a) {
    int j = a.work0 + int(blockIdx.x * blockDim.x + threadIdx.x);
    if (j >= a.work0 + a.nwork_run) return;
    ScanWork &w = a.work[j];
    uint64_t bits = a.chunk_off[w.first_chunk + (w.nunits + 255) / 256] - a.chunk_off[w.first_chunk];
    uint64_t bytes = (bits + 7) >> 3;
    w.raw_bytes = uint32_t(bytes);
    a.scan_pad_bytes[j] = uint32_t(((bytes + 63) & ~uint64_t(63)) + 64);  // +64: the packer may touch one word past the end
}
__global__ void k_scan_place(AsmCtx a) {
    int j = a.work0 + int(blockIdx.x * blockDim.x + threadIdx.x);
    if (j >= a.work0 + a.nwork_run) return;
    a.work[j].raw_off = a.scan_raw_off[j];
    a.work[j].no_room = a.scan_raw_off[j + 1] > a.raw_chunks * 64 ? 1u : 0u;
    a.work[j].out_off = 0xFFFFFFFFu;   // not part of a file until k_layout says so
    if (j == a.work0 + a.nwork_run - 1 && a.scan_raw_off[j + 1] > a.raw_chunks * 64) *a.overflow = 1;
}
__device__ static uint32_t scan_header_bytes(const AsmCtx &a, const ScanWork &w, const EncScan &sc) {
    uint32_t hdr = 0;
    if (sc.ntables) { hdr += 4; for (int t = 0; t < sc.ntables; t++) hdr += 17 + a.tables[w.table_base + t].nsym; }
    return hdr + 2 + 2 + 1 + 2 * sc.ncomp + 3;
}
__device__ static uint64_t scan_stuffing(const AsmCtx &, const ScanWork &w) { return w.no_room ? 0u : w.ff_bytes; }
__global__ void k_scan_cost(AsmCtx a) {
    int j = a.work0 + int(blockIdx.x * blockDim.x + threadIdx.x);
    if (j >= a.work0 + a.nwork_run) return;
    const ScanWork &w = a.work[j];
    a.scan_cost[j] = scan_header_bytes(a, w, a.script[w.scan]) + w.raw_bytes + uint32_t(scan_stuffing(a, w));
}

// byte sink that turns a lane's contiguous output run into aligned dword stores (single bytes only at the ragged ends)
struct ByteRun {
    uint8_t *p;
    uint32_t acc; int n;
    __device__ __forceinline__ void begin(uint8_t *dst) { p = dst; acc = 0; n = 0; }
    __device__ __forceinline__ void push(uint32_t b) {
        if (n == 0 && (reinterpret_cast<uintptr_t>(p) & 3)) { *p++ = uint8_t(b); return; }
        acc |= b << (8 * n);
        if (++n == 4) { *reinterpret_cast<uint32_t *>(p) = acc; p += 4; acc = 0; n = 0; }
    }
    __device__ __forceinline__ void finish() { for (int i = 0; i < n; i++) *p++ = uint8_t(acc >> (8 * i)); n = 0; }
};

__device__ __forceinline__ static int raw_byte(const uint32_t *raw, uint64_t byte_index) {
    return int((raw[byte_index >> 2] >> (24 - 8 * int(byte_index & 3))) & 255u);
}
// which scan owns raw chunk c?  scan_raw_off is sorted; scans are 64-byte aligned
__device__ static int owner_scan(const AsmCtx &a, uint64_t c) {
    uint64_t byte = c * 64;
    int lo = 0, hi = a.nwork;  // invariant: scan_raw_off[lo] <= byte < scan_raw_off[hi]
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (a.scan_raw_off[mid] <= byte) lo = mid; else hi = mid; }
    return lo;
}

// 0xFF bytes of a word of the bit string whose first `nvalid` bytes (big-endian order) belong to the scan
__device__ __forceinline__ static uint32_t ff_bytes_of(uint32_t w, uint32_t nvalid) {
    const uint32_t x = ~w;                                   // a byte of x is zero where the scan has 0xFF
    uint32_t y = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    y = ~(y | x | 0x7F7F7F7Fu);                              // 0x80 in every zero byte of x, exactly
    const uint32_t valid = nvalid >= 4 ? 0x80808080u : (~(0xFFFFFFFFu >> (8 * nvalid)) & 0x80808080u);
    return uint32_t(__builtin_popcount(y & valid));
}
// the stuffing of the scans of one stage: one workgroup per scan, one 64-byte chunk per lane and step.  work[].ff_bytes gets the scan's
// total (all that the scan search and k_layout ask for); the chunks' own counts stay in chunk_ff for k_ff_prefix.
__global__ void __launch_bounds__(256) k_ff_count(AsmCtx a) {
    CSH_SHARED uint32_t s_total;
    ScanWork &w = a.work[a.work0 + int(blockIdx.x)];
    CSH_PHASE_LOOP(3) {
        if (phase == 0) { if (threadIdx.x == 0) s_total = 0; continue; }
        if (phase == 2) { if (threadIdx.x == 0) w.ff_bytes = s_total; continue; }
        const uint32_t nbytes = w.no_room ? 0u : w.raw_bytes;
        const uint64_t c0 = w.raw_off >> 6;
        uint32_t mine = 0;
        for (uint32_t c = threadIdx.x; uint64_t(c) * 64 < nbytes; c += blockDim.x) {
            const uint4 *src = reinterpret_cast<const uint4 *>(a.raw + (c0 + c) * 16);
            const uint4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
            const uint32_t ws[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
            const uint32_t left = nbytes - c * 64;
            uint32_t n = 0;
            CSH_UNROLL
            for (int j = 0; j < 16; j++) n += ff_bytes_of(ws[j], left > uint32_t(4 * j) ? left - uint32_t(4 * j) : 0u);
            a.chunk_ff[c0 + c] = n;
            mine += n;
        }
        if (mine) atomicAdd(&s_total, mine);
    }
}
// the scans that made it into a file: chunk_ff becomes, in place, the number of stuffed bytes in front of each chunk within its scan
// (what k_emit_data adds to a chunk's position).  One workgroup per scan; lane t owns a run of consecutive chunks.
__global__ void __launch_bounds__(256) k_ff_prefix(AsmCtx a) {
    CSH_SHARED uint32_t s_part[256];
    const ScanWork &w = a.work[blockIdx.x];
    const bool listed = w.out_off != 0xFFFFFFFFu && !w.no_room;
    const uint32_t nchunks = listed ? (w.raw_bytes + 63u) >> 6 : 0u;
    const uint32_t per = (nchunks + blockDim.x - 1) / blockDim.x;
    const uint32_t lo = min(nchunks, threadIdx.x * per), hi = min(nchunks, lo + per);
    uint32_t *cf = a.chunk_ff + (w.raw_off >> 6);
    CSH_PHASE_LOOP(2) {
        if (phase == 0) {
            uint32_t n = 0;
            for (uint32_t c = lo; c < hi; c++) n += cf[c];
            s_part[threadIdx.x] = n;
            continue;
        }
        if (lo == hi) continue;
        uint32_t acc = 0;
        for (uint32_t t = 0; t < threadIdx.x; t++) acc += s_part[t];
        for (uint32_t c = lo; c < hi; c++) { const uint32_t n = cf[c]; cf[c] = acc; acc += n; }
    }
}

// per image: where every scan lands in the file, and the file size
__global__ void k_layout(AsmCtx a) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.nimg) return;
    uint64_t pos = a.hdr_off[i + 1] - a.hdr_off[i];
    const uint32_t n = a.img_nlist[i];
    for (uint32_t s = 0; s < n; s++) {
        ScanWork &w = a.work[a.img_list[size_t(i) * CSH_LIST_MAX + s]];
        const uint32_t hdr = scan_header_bytes(a, w, a.script[w.scan]);
        w.out_off = uint32_t(pos);
        w.hdr_bytes = hdr;
        pos += hdr + w.raw_bytes + scan_stuffing(a, w);
    }
    pos += 2;  // EOI
    a.img_size[i] = uint32_t(pos);
    a.img_size_pad[i] = uint32_t((pos + 15) & ~uint64_t(15));
}

// frame header copy + EOI (one lane per image) and DHT/SOS (one lane per scan)
__global__ void k_emit_headers(AsmCtx a) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (*a.overflow) return;
    if (a.img_off[a.nimg] > a.out_cap) { if (t == 0) *a.overflow = 2; return; }
    if (t < a.nimg) {
        uint8_t *o = a.out + a.img_off[t];
        const uint8_t *h = a.hdr_pool + a.hdr_off[t];
        uint32_t n = a.hdr_off[t + 1] - a.hdr_off[t];
        for (uint32_t k = 0; k < n; k++) o[k] = h[k];
        o[a.img_size[t] - 2] = 0xFF; o[a.img_size[t] - 1] = 0xD9;
        return;
    }
    int j = t - a.nimg;
    if (j >= a.nwork) return;
    const ScanWork &w = a.work[j];
    if (w.out_off == 0xFFFFFFFFu) return;   // a candidate of the scan search that did not make it into the file
    const EncScan &sc = a.script[w.scan];
    const ImgDesc &im = a.imgs[w.image];
    uint8_t *o = a.out + a.img_off[w.image] + w.out_off;
    if (sc.ntables) {
        int len = 2;
        for (int k = 0; k < sc.ntables; k++) len += 17 + a.tables[w.table_base + k].nsym;
        *o++ = 0xFF; *o++ = 0xC4; *o++ = uint8_t(len >> 8); *o++ = uint8_t(len);
        for (int k = 0; k < sc.ntables; k++) {
            const DevEncTable &T = a.tables[w.table_base + k];
            *o++ = uint8_t(sc.dht_id[k]);
            for (int l = 1; l <= 16; l++) *o++ = T.bits[l];
            for (int v = 0; v < T.nsym; v++) *o++ = T.vals[v];
        }
    }
    int len = 6 + 2 * sc.ncomp;
    *o++ = 0xFF; *o++ = 0xDA; *o++ = uint8_t(len >> 8); *o++ = uint8_t(len); *o++ = uint8_t(sc.ncomp);
    for (int k = 0; k < sc.ncomp; k++) { *o++ = uint8_t(im.comp_id[sc.comp[k]]); *o++ = uint8_t(sc.sos_tdta[k]); }
    *o++ = uint8_t(sc.Ss); *o++ = uint8_t(sc.Se); *o++ = uint8_t((sc.Ah << 4) | sc.Al);
}

// stuffed copy: one lane per 64-byte raw chunk
// four workgroups per scan of a file (blockIdx.y), one 64-byte chunk per lane and step
__global__ void __launch_bounds__(256) k_emit_data(AsmCtx a) {
    if (*a.overflow) return;
    const ScanWork &w = a.work[blockIdx.x];
    if (w.out_off == 0xFFFFFFFFu || w.no_room) return;
    const uint64_t c0 = w.raw_off >> 6;
    uint8_t *dst = a.out + a.img_off[w.image] + w.out_off + w.hdr_bytes;
    for (uint32_t c = blockIdx.y * blockDim.x + threadIdx.x; uint64_t(c) * 64 < w.raw_bytes; c += gridDim.y * blockDim.x) {
        const uint32_t rel = c * 64;
        ByteRun o; o.begin(dst + rel + a.chunk_ff[c0 + c]);
        // the whole chunk first (four loads in flight at once), then the stores back to back: stores of one lane that are spread
        // over several memory round trips reach HBM as separate 32-byte sector writes instead of merging in the L2
        const uint4 *src = reinterpret_cast<const uint4 *>(a.raw + (c0 + c) * 16);
        const uint4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
        const uint32_t ws[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
        const uint32_t nbytes = w.raw_bytes - rel < 64 ? w.raw_bytes - rel : 64u;
        CSH_UNROLL
        for (int j = 0; j < 16; j++) {
            CSH_UNROLL
            for (int i = 0; i < 4; i++) {
                if (uint32_t(4 * j + i) < nbytes) {
                    uint32_t b = (ws[j] >> (24 - 8 * i)) & 255u;
                    o.push(b);
                    if (b == 0xFFu) o.push(0u);
                }
            }
        }
        o.finish();
    }
}

void launch_scan_sizes(hipStream_t st, const AsmCtx &a) { if (a.nwork_run) CSH_LAUNCH(k_scan_sizes, dim3((a.nwork_run + 255) / 256), dim3(256), st, a); }
void launch_scan_place(hipStream_t st, const AsmCtx &a) { if (a.nwork_run) CSH_LAUNCH(k_scan_place, dim3((a.nwork_run + 255) / 256), dim3(256), st, a); }
void launch_scan_cost(hipStream_t st, const AsmCtx &a) { if (a.nwork_run) CSH_LAUNCH(k_scan_cost, dim3((a.nwork_run + 255) / 256), dim3(256), st, a); }
void launch_ff_count(hipStream_t st, const AsmCtx &a) { if (a.nwork_run) CSH_LAUNCH_PHASED(k_ff_count, 3, dim3(unsigned(a.nwork_run)), dim3(256), st, a); }
void launch_layout(hipStream_t st, const AsmCtx &a) {
    if (a.nimg) CSH_LAUNCH(k_layout, dim3((a.nimg + 63) / 64), dim3(64), st, a);
    if (a.nwork) CSH_LAUNCH_PHASED(k_ff_prefix, 2, dim3(unsigned(a.nwork)), dim3(256), st, a);
}
void launch_emit(hipStream_t st, const AsmCtx &a) {
    int n = a.nimg + a.nwork;
    if (n) CSH_LAUNCH(k_emit_headers, dim3((n + 63) / 64), dim3(64), st, a);
    if (a.nwork) CSH_LAUNCH(k_emit_data, dim3(unsigned(a.nwork), 4), dim3(256), st, a);
}

}  // namespace csh
