// jpeg_host.hpp -- host-side container logic of the JPEG path: marker parsing (T.81 Annex B),
// quality -> quantisation tables, scan scripts and output header emission.  No sample or coefficient
// arithmetic happens on the host: everything below feeds descriptors to the device pipeline.
//
// Mirrors what libcaesium's jpeg module asks of mozjpeg around the hot path (reference call site
// /root/reference/src/compressor.rs:305; parameters /root/reference/src/compressor.rs:411-446).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace csh {

struct HuffSpec {
    bool present = false;
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    int nvals = 0;
    bool operator==(const HuffSpec &o) const;
};

struct JScan {
    int ncomp = 0;
    int comp_idx[4] = {0}, td[4] = {0}, ta[4] = {0};
    int Ss = 0, Se = 63, Ah = 0, Al = 0;
    size_t data_off = 0, data_len = 0;  // entropy-coded segment inside the file
    bool has_marker = false;            // an 0xFF followed by anything but 0x00 lies inside the segment (RSTn, or garbage)
    HuffSpec dc[4], ac[4];              // tables in force at SOS
};

struct JComp {
    int id = 0, h = 1, v = 1, tq = 0;
    int comp_w = 0, comp_h = 0, real_bw = 0, real_bh = 0, bw = 0, bh = 0;
};

struct JpegInfo {
    int width = 0, height = 0, ncomp = 0;
    bool progressive = false;
    int hmax = 1, vmax = 1, mcus_x = 0, mcus_y = 0;
    int restart_interval = 0;
    JComp comp[4];
    uint16_t qt[4][64] = {{0}};  // natural order
    bool qt_present[4] = {false, false, false, false};
    std::vector<JScan> scans;
    std::vector<uint8_t> meta;  // APPn / COM segments, raw, file order
    int adobe_transform = -1;
};

extern const uint8_t kZigZag[64];  // zig-zag index -> natural index

// returns 0 or a CS_ERR_* code with msg filled
int parse_jpeg(const uint8_t *d, size_t n, JpegInfo &out, std::string &msg);
void jpeg_geometry(JpegInfo &j);

// libjpeg quality scaling on mozjpeg base table #3 (pinned by /root/reference/samples/j0.JPG's DQT,
// SURVEY.md 8c.1); natural order
void quality_table(int quality, uint16_t out[64]);

struct OutScan {
    int ncomp; int comp[3];
    int Ss, Se, Ah, Al;
};
// progressive: libjpeg jpeg_simple_progression script; sequential: one interleaved scan
std::vector<OutScan> output_script(int ncomp, bool progressive);

// header bytes that precede the first scan: SOI, JFIF APP0, [metadata], DQT (merged, mozjpeg style), SOFn
// (tables: g.qt[tq] of every table id the components refer to, in order of first use)
std::vector<uint8_t> build_frame_header(const JpegInfo &g, bool progressive, const std::vector<uint8_t> *meta);
}  // namespace csh
