// jpeg_host.cpp -- see jpeg_host.hpp.
#include "jpeg_host.hpp"

#include <cstring>

#include "../../include/caesium_hip.h"

namespace csh {

const uint8_t kZigZag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

bool HuffSpec::operator==(const HuffSpec &o) const {
    if (present != o.present) return false;
    if (!present) return true;
    return nvals == o.nvals && !memcmp(bits, o.bits, 17) && !memcmp(vals, o.vals, nvals);
}

static int cdiv(int a, int b) { return (a + b - 1) / b; }

void jpeg_geometry(JpegInfo &j) {
    j.hmax = j.vmax = 1;
    for (int c = 0; c < j.ncomp; c++) {
        if (j.comp[c].h > j.hmax) j.hmax = j.comp[c].h;
        if (j.comp[c].v > j.vmax) j.vmax = j.comp[c].v;
    }
    j.mcus_x = cdiv(j.width, 8 * j.hmax);
    j.mcus_y = cdiv(j.height, 8 * j.vmax);
    for (int c = 0; c < j.ncomp; c++) {
        JComp &k = j.comp[c];
        k.comp_w = cdiv(j.width * k.h, j.hmax);
        k.comp_h = cdiv(j.height * k.v, j.vmax);
        k.real_bw = cdiv(k.comp_w, 8);
        k.real_bh = cdiv(k.comp_h, 8);
        k.bw = j.mcus_x * k.h;
        k.bh = j.mcus_y * k.v;
    }
}

#define BAD(code, text) do { msg = text; return code; } while (0)

int parse_jpeg(const uint8_t *d, size_t n, JpegInfo &j, std::string &msg) {
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) BAD(CS_ERR_BAD_JPEG, "not a JPEG stream (missing SOI)");
    HuffSpec dc[4], ac[4];
    bool have_sof = false;
    size_t i = 2;
    while (i + 4 <= n) {
        if (d[i] != 0xFF) { i++; continue; }
        int m = d[i + 1];
        if (m == 0xFF) { i++; continue; }
        if (m == 0xD9) break;
        if (m == 0x00 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { i += 2; continue; }
        size_t L = (size_t(d[i + 2]) << 8) | d[i + 3];
        if (L < 2 || i + 2 + L > n) BAD(CS_ERR_BAD_JPEG, "truncated marker segment");
        const uint8_t *s = d + i + 4;
        size_t sl = L - 2;
        switch (m) {
        case 0xDB: {
            size_t p = 0;
            while (p < sl) {
                int pq = s[p] >> 4, tq = s[p] & 15;
                p++;
                size_t need = pq ? 128 : 64;
                if (tq > 3 || pq > 1 || p + need > sl) BAD(CS_ERR_BAD_JPEG, "malformed DQT");
                for (int k = 0; k < 64; k++) {
                    int v = pq ? ((s[p] << 8) | s[p + 1]) : s[p];
                    p += pq ? 2 : 1;
                    j.qt[tq][kZigZag[k]] = uint16_t(v);
                }
                j.qt_present[tq] = true;
            }
            break;
        }
        case 0xC0: case 0xC1: case 0xC2: {
            if (have_sof) BAD(CS_ERR_BAD_JPEG, "multiple SOF markers");
            if (sl < 6) BAD(CS_ERR_BAD_JPEG, "malformed SOF");
            j.progressive = (m == 0xC2);
            int prec = s[0];
            j.height = (s[1] << 8) | s[2];
            j.width = (s[3] << 8) | s[4];
            j.ncomp = s[5];
            if (prec != 8) BAD(CS_ERR_JPEG_FEATURE, "only 8-bit sample precision is supported");
            if (j.ncomp < 1 || j.ncomp > 4 || sl < size_t(6 + 3 * j.ncomp)) BAD(CS_ERR_BAD_JPEG, "malformed SOF");
            if (!j.width || !j.height) BAD(CS_ERR_BAD_JPEG, "empty image");
            for (int c = 0; c < j.ncomp; c++) {
                j.comp[c].id = s[6 + 3 * c];
                j.comp[c].h = s[7 + 3 * c] >> 4;
                j.comp[c].v = s[7 + 3 * c] & 15;
                j.comp[c].tq = s[8 + 3 * c];
                if (j.comp[c].h < 1 || j.comp[c].h > 4 || j.comp[c].v < 1 || j.comp[c].v > 4 || j.comp[c].tq > 3)
                    BAD(CS_ERR_BAD_JPEG, "malformed SOF component");
            }
            jpeg_geometry(j);
            have_sof = true;
            break;
        }
        case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
            BAD(CS_ERR_JPEG_FEATURE, "unsupported JPEG process (lossless / hierarchical / arithmetic)");
        case 0xC4: {
            size_t p = 0;
            while (p + 17 <= sl) {
                int tc = s[p] >> 4, th = s[p] & 15;
                if (tc > 1 || th > 3) BAD(CS_ERR_BAD_JPEG, "malformed DHT");
                HuffSpec &h = tc ? ac[th] : dc[th];
                h = HuffSpec();
                int cnt = 0;
                for (int l = 1; l <= 16; l++) { h.bits[l] = s[p + l]; cnt += s[p + l]; }
                p += 17;
                if (cnt > 256 || p + cnt > sl) BAD(CS_ERR_BAD_JPEG, "malformed DHT");
                // Kraft check, so the device LUT builder can trust the lengths
                int code = 0;
                for (int l = 1; l <= 16; l++) { code = (code + h.bits[l]) << 1; if (code > (2 << l)) BAD(CS_ERR_BAD_JPEG, "oversubscribed DHT"); }
                memcpy(h.vals, s + p, cnt);
                if (!tc) for (int i = 0; i < cnt; i++) if (h.vals[i] > 15) BAD(CS_ERR_BAD_JPEG, "DC Huffman symbol out of range");   // libjpeg JERR_BAD_HUFF_TABLE
                h.nvals = cnt;
                h.present = true;
                p += cnt;
            }
            break;
        }
        case 0xDD:
            if (sl >= 2) j.restart_interval = (s[0] << 8) | s[1];
            break;
        case 0xDA: {
            if (!have_sof) BAD(CS_ERR_BAD_JPEG, "SOS before SOF");
            if (sl < 6) BAD(CS_ERR_BAD_JPEG, "malformed SOS");   // before s[0] is read: a two-byte segment at the very end of the file has no body
            int ns = s[0];
            if (ns < 1 || ns > 4 || sl < size_t(1 + 2 * ns + 3)) BAD(CS_ERR_BAD_JPEG, "malformed SOS");
            JScan sc;
            sc.ncomp = ns;
            for (int k = 0; k < ns; k++) {
                int cid = s[1 + 2 * k], ci = -1;
                for (int c = 0; c < j.ncomp; c++) if (j.comp[c].id == cid) ci = c;
                if (ci < 0) BAD(CS_ERR_BAD_JPEG, "SOS names an unknown component");
                for (int q = 0; q < k; q++) if (sc.comp_idx[q] == ci) BAD(CS_ERR_BAD_JPEG, "SOS names a component twice");   // libjpeg JERR_BAD_COMPONENT_ID
                sc.comp_idx[k] = ci;
                sc.td[k] = s[2 + 2 * k] >> 4;
                sc.ta[k] = s[2 + 2 * k] & 15;
            }
            sc.Ss = s[1 + 2 * ns]; sc.Se = s[2 + 2 * ns];
            sc.Ah = s[3 + 2 * ns] >> 4; sc.Al = s[3 + 2 * ns] & 15;
            if (!j.progressive) { sc.Ss = 0; sc.Se = 63; sc.Ah = sc.Al = 0; }
            else {
                if (sc.Ss > sc.Se || sc.Se > 63 || sc.Al > 13 || (sc.Ah != 0 && sc.Al != sc.Ah - 1)) BAD(CS_ERR_BAD_JPEG, "bad progressive parameters");   // libjpeg jdphuff.c: JERR_BAD_PROGRESSION
                if (sc.Ss == 0 && sc.Se != 0) BAD(CS_ERR_BAD_JPEG, "bad progressive DC scan");
                if (sc.Ss != 0 && ns != 1) BAD(CS_ERR_BAD_JPEG, "interleaved progressive AC scan");
            }
            for (int k = 0; k < ns; k++) {   // only the selectors the scan uses are checked (libjpeg: jpeg_make_d_derived_tbl on use)
                const bool need_dc = j.progressive ? (sc.Ss == 0 && sc.Ah == 0) : true, need_ac = j.progressive ? sc.Ss != 0 : true;
                if ((need_dc && sc.td[k] > 3) || (need_ac && sc.ta[k] > 3)) BAD(CS_ERR_BAD_JPEG, "SOS table id out of range");
                sc.td[k] &= 3; sc.ta[k] &= 3;
            }
            for (int t = 0; t < 4; t++) { sc.dc[t] = dc[t]; sc.ac[t] = ac[t]; }
            size_t b = i + 2 + L, e = b;
            while (e + 1 < n) {   // hop from 0xFF to 0xFF (memchr), the data in between cannot end the scan
                const void *f = memchr(d + e, 0xFF, n - 1 - e);
                if (!f) { e = n; break; }
                e = size_t(static_cast<const uint8_t *>(f) - d);
                if (d[e + 1] != 0x00 && d[e + 1] != 0xFF && !(d[e + 1] >= 0xD0 && d[e + 1] <= 0xD7)) break;
                if (d[e + 1] != 0x00) sc.has_marker = true;
                e++;
            }
            if (e + 1 >= n) e = n;
            sc.data_off = b;
            sc.data_len = e - b;
            if (e > b && d[e - 1] == 0xFF) sc.has_marker = true;   // a dangling 0xFF at the end of the data
            j.scans.push_back(sc);
            i = e;
            continue;
        }
        default:
            if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) {
                if (m == 0xEE && sl >= 12 && !memcmp(s, "Adobe", 5)) j.adobe_transform = s[11];
                j.meta.insert(j.meta.end(), d + i, d + i + 2 + L);
            }
        }
        i += 2 + L;
    }
    if (!have_sof) BAD(CS_ERR_BAD_JPEG, "no frame header");
    if (j.scans.empty()) BAD(CS_ERR_BAD_JPEG, "no scan data");
    for (int c = 0; c < j.ncomp; c++)
        if (!j.qt_present[j.comp[c].tq]) BAD(CS_ERR_BAD_JPEG, "missing quantisation table");
    return 0;
}

// mozjpeg base table index 3 (natural order); j0.JPG's DQT equals this at libjpeg scale 98
static const uint16_t kBaseTable3[64] = {16, 16, 16, 18, 25,  37,  56,  85,  16, 17, 20,  27,  34,  40,  53,  75,
                                         16, 20, 24, 31, 43,  62,  91,  135, 18, 27, 31,  40,  53,  74,  106, 156,
                                         25, 34, 43, 53, 69,  94,  131, 189, 37, 40, 62,  74,  94,  124, 169, 238,
                                         56, 53, 91, 106, 131, 169, 226, 311, 85, 75, 135, 156, 189, 238, 311, 418};

void quality_table(int q, uint16_t out[64]) {
    if (q <= 0) q = 1;
    if (q > 100) q = 100;
    int s = q < 50 ? 5000 / q : 200 - 2 * q;
    for (int i = 0; i < 64; i++) {
        long v = (long(kBaseTable3[i]) * s + 50) / 100;
        if (v < 1) v = 1;
        if (v > 32767) v = 32767;
        out[i] = uint16_t(v);
    }
}

std::vector<OutScan> output_script(int ncomp, bool progressive) {
    std::vector<OutScan> v;
    auto all = [&](int ss, int se, int ah, int al) { OutScan s{ncomp, {0, 1, 2}, ss, se, ah, al}; v.push_back(s); };
    auto one = [&](int c, int ss, int se, int ah, int al) { OutScan s{1, {c, 0, 0}, ss, se, ah, al}; v.push_back(s); };
    if (!progressive) { all(0, 63, 0, 0); return v; }
    if (ncomp == 3) {
        all(0, 0, 0, 1); one(0, 1, 5, 0, 2); one(2, 1, 63, 0, 1); one(1, 1, 63, 0, 1); one(0, 6, 63, 0, 2);
        one(0, 1, 63, 2, 1); all(0, 0, 1, 0); one(2, 1, 63, 1, 0); one(1, 1, 63, 1, 0); one(0, 1, 63, 1, 0);
    } else {
        all(0, 0, 0, 1); one(0, 1, 5, 0, 2); one(0, 6, 63, 0, 2); one(0, 1, 63, 2, 1); all(0, 0, 1, 0); one(0, 1, 63, 1, 0);
    }
    return v;
}

static void put2(std::vector<uint8_t> &b, int v) { b.push_back(uint8_t(v >> 8)); b.push_back(uint8_t(v)); }

std::vector<uint8_t> build_frame_header(const JpegInfo &g, bool progressive, const std::vector<uint8_t> *meta) {
    std::vector<uint8_t> b;
    b.push_back(0xFF); b.push_back(0xD8);
    static const uint8_t jfif[] = {0xFF, 0xE0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
    b.insert(b.end(), jfif, jfif + sizeof jfif);
    if (meta) b.insert(b.end(), meta->begin(), meta->end());
    int ids[4], ntab = 0, prec[4], len = 2;
    bool any16 = false;
    for (int c = 0; c < g.ncomp; c++) {
        bool seen = false;
        for (int t = 0; t < ntab; t++) seen |= ids[t] == g.comp[c].tq;
        if (!seen) ids[ntab++] = g.comp[c].tq;
    }
    for (int t = 0; t < ntab; t++) {
        prec[t] = 0;
        for (int k = 0; k < 64; k++) if (g.qt[ids[t]][k] > 255) prec[t] = 1;
        any16 |= prec[t] != 0;
        len += 1 + (prec[t] ? 128 : 64);
    }
    b.push_back(0xFF); b.push_back(0xDB); put2(b, len);  // one DQT segment for all tables (mozjpeg marker style, cf. samples/j0.JPG)
    for (int t = 0; t < ntab; t++) {
        b.push_back(uint8_t((prec[t] << 4) | ids[t]));
        for (int k = 0; k < 64; k++) { int v = g.qt[ids[t]][kZigZag[k]]; if (prec[t]) b.push_back(uint8_t(v >> 8)); b.push_back(uint8_t(v)); }
    }
    bool baseline = !progressive && !any16;
    b.push_back(0xFF); b.push_back(progressive ? 0xC2 : (baseline ? 0xC0 : 0xC1));
    put2(b, 8 + 3 * g.ncomp); b.push_back(8); put2(b, g.height); put2(b, g.width); b.push_back(uint8_t(g.ncomp));
    for (int c = 0; c < g.ncomp; c++) { b.push_back(uint8_t(g.comp[c].id)); b.push_back(uint8_t((g.comp[c].h << 4) | g.comp[c].v)); b.push_back(uint8_t(g.comp[c].tq)); }
    return b;
}

}  // namespace csh
