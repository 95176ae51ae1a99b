// cli.cpp -- see cli.hpp.  Host logic only; every pixel goes through include/caesium_hip.h.
#include "cli.hpp"

#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <future>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <unordered_set>
#include <thread>
#include <tuple>

#include "../../include/caesium_hip.h"

namespace cli {

// ------------------------------------------------------------------------------------------------ sizes
bool parse_bytesize(const std::string &in, uint64_t &out) {
    // bytesize FromStr: <float><optional spaces><unit>, units B, K/KB/KiB ... case-insensitive; plain number = bytes
    std::string s = in;
    size_t i = 0;
    while (i < s.size() && (isdigit((unsigned char)s[i]) || s[i] == '.')) i++;
    if (i == 0) return false;
    double v;
    try { size_t used = 0; v = std::stod(s.substr(0, i), &used); if (used != i) return false; } catch (...) { return false; }
    std::string u = s.substr(i);
    while (!u.empty() && u[0] == ' ') u.erase(0, 1);
    std::string l;
    for (char c : u) l.push_back((char)tolower((unsigned char)c));
    static const std::map<std::string, double> units = {
        {"", 1.0}, {"b", 1.0},
        {"k", 1e3}, {"kb", 1e3}, {"ki", 1024.0}, {"kib", 1024.0},
        {"m", 1e6}, {"mb", 1e6}, {"mi", 1048576.0}, {"mib", 1048576.0},
        {"g", 1e9}, {"gb", 1e9}, {"gi", 1073741824.0}, {"gib", 1073741824.0},
        {"t", 1e12}, {"tb", 1e12}, {"ti", 1099511627776.0}, {"tib", 1099511627776.0},
        {"p", 1e15}, {"pb", 1e15}, {"pi", 1125899906842624.0}, {"pib", 1125899906842624.0}};
    auto f = units.find(l);
    if (f == units.end()) return false;
    out = uint64_t(v * f->second);
    return true;
}

bool parse_min_savings(const std::string &val, MinSavings &out, std::string &err) {
    std::string t = val;
    while (!t.empty() && isspace((unsigned char)t.front())) t.erase(0, 1);
    while (!t.empty() && isspace((unsigned char)t.back())) t.pop_back();
    if (t.empty()) { err = "Value cannot be empty. Use percentage (e.g., '10%'), size with unit (e.g., '100KB', '1MB'), or plain number as bytes"; return false; }
    if (t.back() == '%') {
        std::string p = t.substr(0, t.size() - 1);
        while (!p.empty() && isspace((unsigned char)p.back())) p.pop_back();
        double d;
        try { size_t used = 0; d = std::stod(p, &used); if (used != p.size()) throw 1; } catch (...) { err = "Invalid percentage value: '" + p + "'"; return false; }
        if (!(d >= 0.0 && d <= 100.0)) { err = "Percentage must be between 0 and 100, got " + p; return false; }
        out.percent = true; out.pct = d;
        return true;
    }
    uint64_t b;
    if (!parse_bytesize(t, b)) { err = "Invalid size format: '" + val + "'. Use percentage (e.g., '10%'), size with unit (e.g., '100KB', '1MB'), or plain number as bytes"; return false; }
    out.percent = false; out.bytes = b;
    return true;
}

std::string format_bytesize(uint64_t n) {
    if (n < 1024) return std::to_string(n) + " B";
    static const char *u[] = {"KiB", "MiB", "GiB", "TiB", "PiB", "EiB"};
    double v = double(n) / 1024.0;
    int i = 0;
    while (v >= 1024.0 && i < 5) { v /= 1024.0; i++; }
    char buf[64];
    snprintf(buf, sizeof buf, "%.1f %s", v, u[i]);
    return buf;
}

// ------------------------------------------------------------------------------------------------ flags
static bool parse_range(const std::string &v, long lo, long hi, const char *name, long &out, std::string &err) {
    char *end = nullptr;
    long x = strtol(v.c_str(), &end, 10);
    if (v.empty() || *end) { err = "'" + v + "' is not a valid number"; return false; }
    if (x < lo || x > hi) { err = std::string(name) + " must be between " + std::to_string(lo) + " and " + std::to_string(hi) + ", but got " + std::to_string(x); return false; }
    out = x;
    return true;
}

std::string usage() {
    return "Usage: caesiumclt [OPTIONS] <--quality <QUALITY>|--lossless|--max-size <MAX_SIZE>> <--output <OUTPUT>|--same-folder-as-input> [FILES]...\n\n"
           "Options:\n"
           "  -q, --quality <QUALITY>          Compression quality [0-100], higher values mean better quality\n"
           "      --lossless                   Use lossless compression\n"
           "      --max-size <MAX_SIZE>        Target maximum file size in bytes or human-readable format (e.g., 100KB, 0.5MB)\n"
           "      --width <WIDTH> / --height <HEIGHT> / --long-edge <PX> / --short-edge <PX> / --no-upscale\n"
           "  -o, --output <OUTPUT>            Output directory path\n"
           "      --same-folder-as-input       Use input file's directory as output\n"
           "      --format <FORMAT>            jpeg|png|gif|webp|tiff|original [default: original]\n"
           "      --png-opt-level <N>          [0-6] [default: 3]\n"
           "      --jpeg-chroma-subsampling <S>  4:4:4|4:2:2|4:2:0|4:1:1|auto [default: auto]\n"
           "      --jpeg-baseline  --zopfli  -e, --exif  --keep-dates  --strip-icc  --suffix <SUFFIX>\n"
           "  -R, --recursive  -S, --keep-structure  -d, --dry-run  --threads <N>  --check-extension-only\n"
           "  -O, --overwrite <all|never|bigger>  --min-savings <10%|100KB|N>\n"
           "  -Q, --quiet | --verbose <0-3> | --json\n"
           "      --gpus <N>                   (caesium-hip) devices to shard the batch over [default: 1]\n"
           "  -h, --help  -V, --version\n";
}

// clap's short-option forms: clusters (-RS, -Rd), attached values (-q80, -q=80, -Oall, -o=dir), a value-taking flag inside a cluster
// takes the rest of the token (-RSq80); a bare '-' and everything behind '--' are positionals
static std::vector<std::string> expand_short_options(const std::vector<std::string> &in) {
    std::vector<std::string> out;
    for (size_t i = 0; i < in.size(); i++) {
        const std::string &t = in[i];
        if (t == "--") { out.insert(out.end(), in.begin() + i, in.end()); break; }
        if (t.size() < 3 || t[0] != '-' || t[1] == '-') { out.push_back(t); continue; }
        for (size_t k = 1; k < t.size(); k++) {
            const char c = t[k];
            out.push_back(std::string("-") + c);
            if (c == 'q' || c == 'o' || c == 'O') {   // takes a value: the rest of the token is it
                std::string rest = t.substr(k + 1);
                if (!rest.empty() && rest[0] == '=') rest.erase(0, 1);
                if (!rest.empty() || (k + 1 < t.size())) out.push_back(rest);
                break;
            }
        }
    }
    return out;
}

bool parse_args(const std::vector<std::string> &args_in, Options &o, std::string &err) {
    const std::vector<std::string> a = expand_short_options(args_in);
    bool verbose_set = false;
    auto need = [&](size_t &i, const std::string &flag, std::string &val) {
        size_t eq = a[i].find('=');
        if (a[i].rfind("--", 0) == 0 && eq != std::string::npos) { val = a[i].substr(eq + 1); return true; }
        if (i + 1 >= a.size()) { err = "a value is required for '" + flag + "' but none was supplied"; return false; }
        val = a[++i];
        return true;
    };
    for (size_t i = 0; i < a.size(); i++) {
        std::string f = a[i];
        if (f.rfind("--", 0) == 0 && f.find('=') != std::string::npos) f = f.substr(0, f.find('='));
        std::string v; long n;
        if (f == "-h" || f == "--help") o.help = true;
        else if (f == "-V" || f == "--version") o.version = true;
        else if (f == "-q" || f == "--quality") { if (!need(i, f, v) || !parse_range(v, 0, 100, "Quality", n, err)) return false; o.quality = uint32_t(n); }
        else if (f == "--lossless") o.lossless = true;
        else if (f == "--max-size") { uint64_t b; if (!need(i, f, v)) return false; if (!parse_bytesize(v, b)) { err = "Invalid size format: " + v; return false; } o.max_size = size_t(b); }
        else if (f == "--width") { if (!need(i, f, v) || !parse_range(v, 0, 0x7fffffff, "width", n, err)) return false; o.width = uint32_t(n); }
        else if (f == "--height") { if (!need(i, f, v) || !parse_range(v, 0, 0x7fffffff, "height", n, err)) return false; o.height = uint32_t(n); }
        else if (f == "--long-edge") { if (!need(i, f, v) || !parse_range(v, 0, 0x7fffffff, "long-edge", n, err)) return false; o.long_edge = uint32_t(n); }
        else if (f == "--short-edge") { if (!need(i, f, v) || !parse_range(v, 0, 0x7fffffff, "short-edge", n, err)) return false; o.short_edge = uint32_t(n); }
        else if (f == "--no-upscale") o.no_upscale = true;
        else if (f == "-o" || f == "--output") { if (!need(i, f, v)) return false; o.output = fs::path(v); }
        else if (f == "--same-folder-as-input") o.same_folder_as_input = true;
        else if (f == "--format") {
            if (!need(i, f, v)) return false;
            if (v == "jpeg") o.format = Format::Jpeg; else if (v == "png") o.format = Format::Png; else if (v == "gif") o.format = Format::Gif;
            else if (v == "webp") o.format = Format::Webp; else if (v == "tiff") o.format = Format::Tiff; else if (v == "original") o.format = Format::Original;
            else { err = "invalid value '" + v + "' for '--format <FORMAT>'"; return false; }
        }
        else if (f == "--png-opt-level") { if (!need(i, f, v) || !parse_range(v, 0, 6, "PNG optimization level", n, err)) return false; o.png_opt_level = int(n); }
        else if (f == "--jpeg-chroma-subsampling") {
            if (!need(i, f, v)) return false;
            if (v == "4:4:4") o.chroma = 444; else if (v == "4:2:2") o.chroma = 422; else if (v == "4:2:0") o.chroma = 420; else if (v == "4:1:1") o.chroma = 411;
            else if (v == "auto") o.chroma = 0; else { err = "invalid value '" + v + "' for '--jpeg-chroma-subsampling'"; return false; }
        }
        else if (f == "--jpeg-baseline") o.jpeg_baseline = true;
        else if (f == "--zopfli") o.zopfli = true;
        else if (f == "-e" || f == "--exif") o.exif = true;
        else if (f == "--keep-dates") o.keep_dates = true;
        else if (f == "--strip-icc") o.strip_icc = true;
        else if (f == "--suffix") { if (!need(i, f, v)) return false; o.suffix = v; }
        else if (f == "-R" || f == "--recursive") o.recursive = true;
        else if (f == "-S" || f == "--keep-structure") o.keep_structure = true;
        else if (f == "-d" || f == "--dry-run") o.dry_run = true;
        else if (f == "--threads") { if (!need(i, f, v) || !parse_range(v, 0, 0x7fffffff, "threads", n, err)) return false; o.threads = uint32_t(n); }
        else if (f == "--check-extension-only") o.check_extension_only = true;
        else if (f == "-O" || f == "--overwrite") {
            if (!need(i, f, v)) return false;
            if (v == "all") o.overwrite = Overwrite::All; else if (v == "never") o.overwrite = Overwrite::Never; else if (v == "bigger") o.overwrite = Overwrite::Bigger;
            else { err = "invalid value '" + v + "' for '--overwrite <OVERWRITE>'"; return false; }
        }
        else if (f == "--min-savings") { MinSavings m; if (!need(i, f, v) || !parse_min_savings(v, m, err)) return false; o.min_savings = m; }
        else if (f == "-Q" || f == "--quiet") o.quiet = true;
        else if (f == "--verbose") { if (!need(i, f, v) || !parse_range(v, 0, 3, "Verbosity", n, err)) return false; o.verbose = int(n); verbose_set = true; }
        else if (f == "--json") o.json = true;
        else if (f == "--gpus") { if (!need(i, f, v) || !parse_range(v, 1, 64, "gpus", n, err)) return false; o.gpus = int(n); }
        else if (f == "--") { for (size_t j = i + 1; j < a.size(); j++) o.files.push_back(a[j]); break; }
        else if (f.size() > 1 && f[0] == '-') { err = "unexpected argument '" + a[i] + "' found"; return false; }
        else o.files.push_back(a[i]);
    }
    if (o.help || o.version) return true;
    int modes = (o.quality ? 1 : 0) + (o.lossless ? 1 : 0) + (o.max_size ? 1 : 0);
    if (modes == 0) { err = "the following required arguments were not provided:\n  <--quality <QUALITY>|--lossless|--max-size <MAX_SIZE>>"; return false; }
    if (modes > 1) { err = "the argument '--quality <QUALITY>' cannot be used with '--lossless' / '--max-size <MAX_SIZE>'"; return false; }
    int dest = (o.output ? 1 : 0) + (o.same_folder_as_input ? 1 : 0);
    if (dest == 0) { err = "the following required arguments were not provided:\n  <--output <OUTPUT>|--same-folder-as-input>"; return false; }
    if (dest > 1) { err = "the argument '--output <OUTPUT>' cannot be used with '--same-folder-as-input'"; return false; }
    if ((o.width || o.height) && (o.long_edge || o.short_edge)) { err = "the argument '--width/--height' cannot be used with '--long-edge/--short-edge'"; return false; }
    if (o.long_edge && o.short_edge) { err = "the argument '--long-edge <LONG_EDGE>' cannot be used with '--short-edge <SHORT_EDGE>'"; return false; }
    if ((o.quiet ? 1 : 0) + (verbose_set ? 1 : 0) + (o.json ? 1 : 0) > 1) { err = "the arguments '--quiet', '--verbose <VERBOSE>' and '--json' cannot be used together"; return false; }
    return true;
}

// ------------------------------------------------------------------------------------------------ scan
static std::string lower(std::string s) { for (char &c : s) c = (char)tolower((unsigned char)c); return s; }
bool has_supported_extension(const fs::path &p) {
    std::string e = lower(p.extension().string());
    return e == ".jpg" || e == ".jpeg" || e == ".png" || e == ".webp" || e == ".gif";
}
bool is_filetype_supported(const fs::path &p) {
    unsigned char b[16];
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) return false;
    size_t n = fread(b, 1, 16, f);
    fclose(f);
    if (n < 16) return false;  // read_exact(16) in the reference
    if (b[0] == 0xFF && b[1] == 0xD8 && b[2] == 0xFF) return true;
    if (!memcmp(b, "\x89PNG", 4)) return true;
    if (!memcmp(b, "RIFF", 4) && !memcmp(b + 8, "WEBP", 4)) return true;
    if (!memcmp(b, "GIF8", 4)) return true;
    return false;
}
// std::path::absolute (Unix): prepend the cwd to relative paths, drop `.` components and repeated separators, keep `..`
static bool absolute_rs(const fs::path &p, fs::path &out) {
    if (p.empty()) return false;
    std::error_code ec;
    fs::path full = p.is_absolute() ? p : fs::current_path(ec) / p;
    if (ec) return false;
    out.clear();
    for (const fs::path &c : full) if (c != "." && !c.empty()) out /= c;
    return true;
}
static std::vector<fs::path> components(const fs::path &p) { return std::vector<fs::path>(p.begin(), p.end()); }
static bool has_parent(const fs::path &p) {  // Rust Path::parent().is_some()
    if (p.empty()) return false;
    return p != p.root_path();
}

std::optional<fs::path> compute_base_folder(const std::optional<fs::path> &bf, const fs::path &new_path) {
    if (!bf) { if (!has_parent(new_path)) return std::nullopt; return new_path.parent_path(); }
    const fs::path &base = *bf;
    if (!has_parent(base)) return base;
    fs::path npf = new_path;
    std::error_code ec;
    if (fs::is_regular_file(new_path, ec)) npf = new_path.parent_path();
    auto bc = components(base), nc = components(npf);
    fs::path folder;
    for (size_t i = 0; i < bc.size(); i++) {
        if (i < nc.size() && nc[i] == bc[i]) folder /= bc[i]; else break;
    }
    return folder;
}

namespace { void parallel_for(size_t n, size_t threads, const std::function<void(size_t)> &fn); }
void scan_files(const std::vector<std::string> &args, bool recursive, bool ext_only, std::optional<fs::path> &base, std::vector<fs::path> &files) {
    base.reset();
    auto valid = [&](const fs::path &p) { return ext_only ? has_supported_extension(p) : is_filetype_supported(p); };
    auto add = [&](const fs::path &p) {
        std::error_code ec;
        if (!fs::exists(p, ec)) return;
        fs::path ap;
        if (!absolute_rs(p, ap)) return;
        auto nb = compute_base_folder(base, ap);
        if (!nb) return;
        base = nb;
        files.push_back(p);
    };
    for (const std::string &arg : args) {
        fs::path in(arg);
        std::error_code ec;
        if (fs::exists(in, ec) && fs::is_directory(in, ec)) {
            std::vector<fs::path> found;
            if (recursive) { for (auto it = fs::recursive_directory_iterator(in, fs::directory_options::skip_permission_denied, ec); it != fs::recursive_directory_iterator(); it.increment(ec)) if (it->is_regular_file(ec) && !it->is_symlink(ec)) found.push_back(it->path()); }
            else { for (auto it = fs::directory_iterator(in, fs::directory_options::skip_permission_denied, ec); it != fs::directory_iterator(); it.increment(ec)) if (it->is_regular_file(ec) && !it->is_symlink(ec)) found.push_back(it->path()); }
            std::sort(found.begin(), found.end());
            // (the type check opens every file: on a few threads -- 10 000 files took 80 ms one after the other -- the order stays the sorted one)
            std::vector<char> ok(found.size(), 0);
            parallel_for(found.size(), std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency())), [&](size_t k) { ok[k] = valid(found[k]) ? 1 : 0; });
            for (size_t k = 0; k < found.size(); k++) if (ok[k]) add(found[k]);
        } else if (fs::is_regular_file(in, ec) && valid(in)) add(in);
    }
}

// ------------------------------------------------------------------------------------------------ output path
bool compute_output_full_path(const fs::path &output_directory, const fs::path &input, const fs::path &base_directory, bool keep_structure,
                              const std::string &suffix, Format format, bool same_folder, fs::path &dir_out, std::string &name_out) {
    std::string ext;
    switch (format) {
    case Format::Jpeg: ext = "jpg"; break; case Format::Png: ext = "png"; break; case Format::Webp: ext = "webp"; break;
    case Format::Tiff: ext = "tiff"; break; case Format::Gif: ext = "gif"; break;
    case Format::Original: ext = input.extension().string(); if (!ext.empty()) ext.erase(0, 1); break;
    }
    name_out = input.stem().string() + suffix;
    if (!ext.empty()) name_out += "." + ext;
    if (!keep_structure) { dir_out = output_directory; return true; }
    fs::path parent = input.parent_path();
    std::error_code ec;
    if (!has_parent(input) || parent.empty() || !fs::exists(parent, ec)) return false;   // Path::new("").exists() is false
    if (!absolute_rs(fs::path(parent), parent)) return false;
    if (same_folder) { dir_out = parent; return true; }
    fs::path prefix;
    if (!base_directory.empty()) {
        auto pc = components(parent), bc = components(base_directory);
        if (bc.size() > pc.size()) return false;
        for (size_t i = 0; i < bc.size(); i++) if (pc[i] != bc[i]) return false;
        for (size_t i = bc.size(); i < pc.size(); i++) prefix /= pc[i];
    } else {
        std::string s = parent.string();
        s.erase(std::remove(s.begin(), s.end(), ':'), s.end());
        prefix = fs::path(s);   // absolute: join() below replaces output_directory, as PathBuf::join does
    }
    dir_out = output_directory / prefix;
    return true;
}

// ------------------------------------------------------------------------------------------------ dimensions
static uint32_t be16(const uint8_t *p) { return (uint32_t(p[0]) << 8) | p[1]; }
static uint32_t rd16(const uint8_t *p, bool le) { return le ? (p[0] | (uint32_t(p[1]) << 8)) : be16(p); }
static uint32_t rd32(const uint8_t *p, bool le) { return le ? (p[0] | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24)) : ((uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]); }

bool probe_dimensions(const std::vector<uint8_t> &b, bool keep_metadata, size_t &w, size_t &h) {
    const size_t n = b.size();
    const uint8_t *d = b.data();
    if (n >= 24 && !memcmp(d, "\x89PNG\r\n\x1a\n", 8)) { w = rd32(d + 16, false); h = rd32(d + 20, false); return true; }
    if (n >= 10 && !memcmp(d, "GIF8", 4)) { w = d[6] | (d[7] << 8); h = d[8] | (d[9] << 8); return true; }
    if (n >= 30 && !memcmp(d, "RIFF", 4) && !memcmp(d + 8, "WEBP", 4)) {
        if (!memcmp(d + 12, "VP8 ", 4)) { w = (d[26] | (d[27] << 8)) & 0x3FFF; h = (d[28] | (d[29] << 8)) & 0x3FFF; return true; }
        if (!memcmp(d + 12, "VP8L", 4)) { uint32_t v = rd32(d + 21, true); w = (v & 0x3FFF) + 1; h = ((v >> 14) & 0x3FFF) + 1; return true; }
        if (!memcmp(d + 12, "VP8X", 4)) { w = (d[24] | (d[25] << 8) | (d[26] << 16)) + 1; h = (d[27] | (d[28] << 8) | (d[29] << 16)) + 1; return true; }
        return false;
    }
    if (n >= 4 && d[0] == 0xFF && d[1] == 0xD8) {
        int orientation = 1;
        bool have = false;
        for (size_t i = 2; i + 4 <= n;) {
            if (d[i] != 0xFF) { i++; continue; }
            int m = d[i + 1];
            if (m == 0xFF) { i++; continue; }
            if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { i += 2; continue; }
            size_t L = be16(d + i + 2);
            if (L < 2 || i + 2 + L > n) break;
            if (m >= 0xC0 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC && L >= 7) { h = be16(d + i + 5); w = be16(d + i + 7); have = true; break; }
            if (m == 0xE1 && keep_metadata && L >= 16 && !memcmp(d + i + 4, "Exif\0\0", 6)) {
                const uint8_t *t = d + i + 10; size_t tl = L - 8;
                bool le = t[0] == 'I';
                if (tl >= 8 && (le || t[0] == 'M')) {
                    uint32_t ifd = rd32(t + 4, le);
                    if (uint64_t(ifd) + 2 <= tl) {   // 64-bit bounds: an offset near 2^32 must not wrap back into range
                        uint32_t cnt = rd16(t + ifd, le);
                        for (uint32_t k = 0; k < cnt && uint64_t(ifd) + 2 + 12ull * (k + 1) <= tl; k++) {
                            const uint8_t *e = t + ifd + 2 + 12 * k;
                            if (rd16(e, le) == 0x0112) orientation = int(rd16(e + 8, le));
                        }
                    }
                }
            }
            if (m == 0xDA) break;
            i += 2 + L;
        }
        if (!have) return false;
        if (orientation >= 5 && orientation <= 8) std::swap(w, h);
        return true;
    }
    return false;
}

// ------------------------------------------------------------------------------------------------ reporting
static std::string json_escape(const std::string &s) {
    std::string o = "\"";
    for (unsigned char c : s) {
        switch (c) {
        case '"': o += "\\\""; break; case '\\': o += "\\\\"; break; case '\n': o += "\\n"; break; case '\r': o += "\\r"; break;
        case '\t': o += "\\t"; break; case '\b': o += "\\b"; break; case '\f': o += "\\f"; break;
        default: if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; } else o.push_back((char)c);
        }
    }
    return o + "\"";
}
static std::string json_f64(double v) {  // serde_json: shortest round-trip, always with a fractional part
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, v);
    std::string s(buf, r.ptr);
    if (s.find('.') == std::string::npos && s.find('e') == std::string::npos && s.find("inf") == std::string::npos && s.find("nan") == std::string::npos) s += ".0";
    return s;
}
struct Stats { uint64_t orig = 0, comp = 0; size_t success = 0, skipped = 0, errors = 0; };
static Stats fold(const std::vector<Result> &r) {
    Stats s;
    for (const Result &x : r) {
        s.orig += x.original_size; s.comp += x.compressed_size;
        if (x.status == Status::Success) s.success++; else if (x.status == Status::Skipped) s.skipped++; else s.errors++;
    }
    return s;
}
static const char *status_lower(Status s) { return s == Status::Success ? "success" : s == Status::Skipped ? "skipped" : "error"; }

std::string build_json(const std::vector<Result> &results, bool dry_run, const char *error) {
    Stats s = fold(results);
    int64_t saved = int64_t(s.orig) - int64_t(s.comp);
    double pct = s.orig > 0 ? (double(saved) / double(s.orig)) * 100.0 : 0.0;
    std::string o = "{\"version\":\"1.0.0\",\"dry_run\":";
    o += dry_run ? "true" : "false";
    o += ",\"error\":";
    o += error ? json_escape(error) : "null";
    o += ",\"files\":[";
    for (size_t i = 0; i < results.size(); i++) {
        const Result &r = results[i];
        if (i) o += ",";
        o += "{\"original_path\":" + json_escape(r.original_path) + ",\"output_path\":" + json_escape(r.output_path) + ",\"original_size\":" + std::to_string(r.original_size) +
             ",\"compressed_size\":" + std::to_string(r.compressed_size) + ",\"status\":\"" + status_lower(r.status) + "\",\"message\":" + json_escape(r.message) + "}";
    }
    o += "],\"summary\":{\"total_files\":" + std::to_string(results.size()) + ",\"success\":" + std::to_string(s.success) + ",\"skipped\":" + std::to_string(s.skipped) +
         ",\"errors\":" + std::to_string(s.errors) + ",\"original_size\":" + std::to_string(s.orig) + ",\"compressed_size\":" + std::to_string(s.comp) +
         ",\"savings_bytes\":" + std::to_string(saved) + ",\"savings_percent\":" + json_f64(pct) + "}}";
    return o;
}

static std::string paint(const std::string &s, const char *code, bool color) { return color ? std::string("\x1b[") + code + "m" + s + "\x1b[0m" : s; }
static void savings_strings(int64_t saved, double pct, bool color, std::string &sz, std::string &pc) {
    char b[64];
    uint64_t a = uint64_t(saved < 0 ? -saved : saved);
    if (saved >= 0) { sz = paint("-" + format_bytesize(a), "32", color); snprintf(b, sizeof b, "-%.2f%%", pct); pc = paint(b, "32", color); }
    else { sz = paint("+" + format_bytesize(a), "31", color); snprintf(b, sizeof b, "+%.2f%%", -pct); pc = paint(b, "31", color); }
}
std::string build_recap(const std::vector<Result> &results, int verbose, bool color) {
    std::string o;
    if (results.empty()) return o;
    Stats s = fold(results);
    if (verbose > 1)
        for (const Result &r : results) {
            if (verbose < 3 && r.status == Status::Success) continue;
            int64_t saved = int64_t(r.original_size) - int64_t(r.compressed_size);
            double pct = r.original_size > 0 ? (double(saved) / double(r.original_size)) * 100.0 : 0.0;
            std::string sz, pc;
            savings_strings(saved, pct, color, sz, pc);
            const char *code = r.status == Status::Success ? "32" : r.status == Status::Skipped ? "33" : "31";
            const char *name = r.status == Status::Success ? "Success" : r.status == Status::Skipped ? "Skipped" : "Error";
            o += "[" + paint(name, code, color) + "] " + r.original_path + " -> " + r.output_path + "\n" + format_bytesize(r.original_size) + " -> " +
                 format_bytesize(r.compressed_size) + " [" + sz + " | " + pc + "]\n";
            if (!r.message.empty()) o += paint(r.message, code, color) + "\n";
            o += "\n";
        }
    if (verbose > 0) {
        int64_t saved = int64_t(s.orig) - int64_t(s.comp);
        double pct = s.orig > 0 ? (double(saved) / double(s.orig)) * 100.0 : 0.0;
        std::string sz, pc;
        savings_strings(saved, pct, color, sz, pc);
        o += "Compressed " + std::to_string(results.size()) + " files (" + paint(std::to_string(s.success), "32", color) + " success, " + paint(std::to_string(s.skipped), "33", color) +
             " skipped, " + paint(std::to_string(s.errors), "31", color) + " errors)\n" + format_bytesize(s.orig) + " -> " + format_bytesize(s.comp) + " [" + sz + " | " + pc + "]\n";
    }
    return o;
}

size_t parallelism_count(uint32_t requested, size_t available) { return requested == 0 ? available : std::min<size_t>(requested, available); }

// ------------------------------------------------------------------------------------------------ the run
namespace {
const uint64_t kMaxFileSize = 500ull * 1024 * 1024;

struct Job {
    fs::path input, output;
    struct stat st {};
    std::vector<uint8_t> data;
    CCSParameters params{};
    bool engine = false;  // reached the engine stage
    CByteArray result{nullptr, 0};   // owned: the engine's output buffer, released after the write (no copy)
    bool ok = false;
    std::string engine_msg;
};

void parallel_for(size_t n, size_t threads, const std::function<void(size_t)> &fn) {
    if (threads <= 1 || n <= 1) { for (size_t i = 0; i < n; i++) fn(i); return; }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (size_t t = 0; t < std::min(threads, n); t++) pool.emplace_back([&] { for (size_t i; (i = next++) < n;) fn(i); });
    for (auto &t : pool) t.join();
}

bool read_file(const fs::path &p, std::vector<uint8_t> &out) {
    std::ifstream f(p, std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end);
    std::streamoff n = f.tellg();
    f.seekg(0);
    out.resize(size_t(n > 0 ? n : 0));
    if (n > 0) f.read(reinterpret_cast<char *>(out.data()), n);
    return bool(f) || f.eof();
}

// compressor.rs:411-446 + 503-536
bool build_parameters(const Options &o, const std::vector<uint8_t> &buf, CCSParameters &p, std::string &err) {
    cs_default_parameters(&p);
    uint32_t q = o.quality.value_or(80);
    p.jpeg_quality = p.png_quality = p.webp_quality = q;
    p.gif_quality = o.lossless ? 100 : (q == 0 ? 1 : q);
    p.jpeg_preserve_icc = !o.strip_icc;
    p.jpeg_optimize = p.png_optimize = p.webp_lossless = o.lossless;
    p.keep_metadata = o.exif;
    p.jpeg_chroma_subsampling = uint32_t(o.chroma);
    p.jpeg_progressive = !o.jpeg_baseline;
    p.png_optimization_level = uint32_t(o.png_opt_level);
    p.png_force_zopfli = o.zopfli;
    if (o.width || o.height || o.long_edge || o.short_edge) {
        size_t w, h;
        if (!probe_dimensions(buf, o.exif, w, h)) { err = "could not read the image size"; return false; }
        if (o.width || o.height) { p.width = o.width.value_or(0); p.height = o.height.value_or(0); }
        else if (o.long_edge) { if (w > h) p.width = *o.long_edge; else p.height = *o.long_edge; }
        else if (o.short_edge) { if (w < h) p.width = *o.short_edge; else p.height = *o.short_edge; }
        if (o.no_upscale && (p.width >= w || p.height >= h)) { p.width = 0; p.height = 0; }
    }
    return true;
}
uint32_t map_format(Format f) {
    switch (f) { case Format::Jpeg: return CS_TYPE_JPEG; case Format::Png: return CS_TYPE_PNG; case Format::Gif: return CS_TYPE_GIF;
                 case Format::Webp: return CS_TYPE_WEBP; case Format::Tiff: return CS_TYPE_TIFF; default: return CS_TYPE_UNKN; }
}
}  // namespace

int run(const Options &o) {
    if (o.files.empty()) {
        if (o.json) printf("%s\n", build_json({}, o.dry_run, "No files to compress").c_str()); else fprintf(stderr, "No files to compress\n");
        return 0;
    }
    const bool quiet = o.quiet || o.verbose == 0;
    const int verbose = quiet ? 0 : o.verbose;
    const size_t threads = parallelism_count(o.threads, std::max(1u, std::thread::hardware_concurrency()));
    const bool trace = getenv("CSH_TRACE") != nullptr;   // wall-clock of the stages, on stderr
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t_start = now();
    // the runtime and the device context start while the tree is scanned and read (a cold process pays ~0.15 s for them in front of its first kernel)
    std::vector<std::thread> warm;
    const char *wu = getenv("CSH_CLI_WARMUP");
    if (!o.dry_run && !(wu && !strcmp(wu, "0"))) for (int d = 0; d < (getenv("CSH_CLI_SAME_DEVICE") ? 1 : std::max(1, o.gpus)); d++) warm.emplace_back([d] { csh_warmup(d); });
    struct JoinWarm { std::vector<std::thread> &t; ~JoinWarm() { for (auto &x : t) if (x.joinable()) x.join(); } } join_warm{warm};
    std::optional<fs::path> base;
    std::vector<fs::path> files;
    scan_files(o.files, o.recursive, o.check_extension_only, base, files);
    const auto t_scan = now();
    if (!base) {
        const char *m = "Unable to compute the base path for the files.";
        if (o.json) printf("%s\n", build_json({}, o.dry_run, m).c_str()); else fprintf(stderr, "%s\n", m);
        return 255;  // exit(-1)
    }
    std::vector<Result> results(files.size());
    std::vector<Job> jobs(files.size());
    const std::string suffix = o.suffix.value_or("");

    // The files go through the three stages a WINDOW at a time (at most CSH_CLI_WINDOW files, default 1024, or 8 GiB of input): what is held in
    // memory is a window's inputs and outputs, not the tree's -- the reference works file by file (compressor.rs:81-100) and takes trees of any size.
    std::vector<std::pair<size_t, size_t>> windows;
    {
        const size_t max_files = getenv("CSH_CLI_WINDOW") ? std::max<size_t>(1, size_t(atol(getenv("CSH_CLI_WINDOW")))) : 1024;   // (round 5, 10 000 x 1080p on /dev/shm: 2.1-2.4 s at 512-1024 files per window, 2.7-2.9 s at 4096: the first window is read before anything else happens, and a window's end is a join)
        const uint64_t max_bytes = uint64_t(8) << 30;
        size_t b0 = 0;
        uint64_t bytes = 0;
        for (size_t i = 0; i < files.size(); i++) {
            struct stat st;
            const uint64_t sz = stat(files[i].c_str(), &st) == 0 ? uint64_t(st.st_size) : 0;
            if (i > b0 && (i - b0 >= max_files || bytes + sz > max_bytes)) { windows.emplace_back(b0, i); b0 = i; bytes = 0; }
            bytes += sz;
        }
        windows.emplace_back(b0, files.size());
    }
    double ms_read = 0, ms_engine = 0, ms_write = 0;
    // ---- stage 1 (host, parallel): everything of perform_compression that precedes the engine call.  The next window is read while this one is in the
    // engine and being written (two windows in memory at most)
    std::atomic<long long> us_read{0};
    auto read_window = [&](size_t w0, size_t w1) {
    const auto t_win = now();
    parallel_for(w1 - w0, threads, [&](size_t k) {
        const size_t i = w0 + k;
        Result &r = results[i];
        Job &j = jobs[i];
        j.input = files[i];
        r.original_path = files[i].string();
        if (stat(files[i].c_str(), &j.st) != 0) { r.message = "Error reading file metadata"; return; }
        uint64_t size = uint64_t(j.st.st_size);
        if (size > kMaxFileSize) { r.message = "File exceeds 500Mb, skipping."; r.status = Status::Skipped; return; }
        r.original_size = size;
        fs::path outdir_in;
        if (o.same_folder_as_input) outdir_in = files[i].parent_path(); else outdir_in = *o.output;
        fs::path dir; std::string name;
        std::error_code ec;
        bool same = o.same_folder_as_input || outdir_in == *base;
        if (!compute_output_full_path(outdir_in, files[i], *base, o.keep_structure, suffix, o.format, same, dir, name)) { r.message = "Error setting up output path"; return; }
        if (!o.dry_run && !fs::exists(dir, ec) && !fs::create_directories(dir, ec) && !fs::exists(dir, ec)) { r.message = "Error setting up output path"; return; }
        j.output = dir / name;
        r.output_path = j.output.string();
        if (o.overwrite == Overwrite::Never && fs::exists(j.output, ec)) {
            r.status = Status::Skipped; r.compressed_size = size; r.message = "File already exists, skipped due overwrite policy"; return;
        }
        if (o.dry_run) { r.status = Status::Success; r.compressed_size = size; return; }
        if (!read_file(files[i], j.data)) { r.message = "Error reading input file"; return; }
        std::string perr;
        if (!build_parameters(o, j.data, j.params, perr)) { r.message = "Error building compression parameters: " + perr; return; }
        j.engine = true;
    });
    us_read += (long long)(ms(t_win, now()) * 1000.0);
    };
    // ---- stage 3 (host): the rest of perform_compression for one file -- savings and overwrite policy, the write.  It runs per device batch, as soon as
    // the batch comes back, on the batch's own worker (a few threads): the writes of one batch overlap the kernels of the next ones
    std::atomic<long long> us_write{0};
    auto finish_job = [&](size_t i) {
        Job &j = jobs[i];
        Result &r = results[i];
        if (!j.engine) return;
        if (!j.ok) { r.message = j.engine_msg; return; }
        const uint64_t orig = r.original_size, outsz = j.result.length;
        struct Release { CByteArray *b; ~Release() { cs_free_bytes(b); } } release{&j.result};
        if (o.min_savings && orig != 0) {
            uint64_t saved = orig > outsz ? orig - outsz : 0;
            char b[160];
            if (o.min_savings->percent) {
                double sp = (double(saved) / double(orig)) * 100.0;
                if (sp < o.min_savings->pct) { snprintf(b, sizeof b, "Insufficient savings: %.2f%% < %.2f%%, skipped", sp, o.min_savings->pct); r.status = Status::Skipped; r.compressed_size = orig; r.message = b; return; }
            } else if (saved < o.min_savings->bytes) {
                r.status = Status::Skipped; r.compressed_size = orig;
                r.message = "Insufficient savings: " + format_bytesize(saved) + " < " + format_bytesize(o.min_savings->bytes) + ", skipped";
                return;
            }
        }
        std::error_code ec;
        if (o.overwrite == Overwrite::Bigger && fs::exists(j.output, ec)) {
            uint64_t existing = fs::file_size(j.output, ec);
            if (ec) r.message = "Error reading existing file metadata";
            else if (existing <= outsz) { r.status = Status::Skipped; r.compressed_size = orig; r.message = "File already exists, skipped due overwrite policy"; return; }
        }
        FILE *f = fopen(j.output.c_str(), "wb");
        if (!f) { r.message = "Error creating output file"; return; }
        bool wrote = fwrite(j.result.data, 1, j.result.length, f) == j.result.length;
        if (wrote && o.keep_dates) {
            fflush(f);
            struct timespec ts[2] = {j.st.st_atim, j.st.st_mtim};
            if (futimens(fileno(f), ts) != 0) { fclose(f); r.message = "Error preserving file times"; return; }
        }
        fclose(f);
        if (!wrote) { r.message = "Error writing output file"; return; }
        r.status = Status::Success;
        r.compressed_size = outsz;
    };
    // Ordering: the read-ahead stats, checks the overwrite policy of and reads window N + 1 while window N is written.  That is only sound when no output
    // of one file is the input (or the overwrite-policy target) of another: `--format` or `--suffix` writing into the input tree can make a.png -> a.jpg land
    // on the input a.jpg.  (The reference's par_iter has the same race between its threads; a sequential reader has not.)  With more than one window,
    // every file's output path is computed up front; if one names another file's input the windows run strictly one after the other.
    bool read_ahead = windows.size() > 1;
    if (read_ahead) {
        std::error_code cwd_ec;
        const fs::path cwd = fs::current_path(cwd_ec);   // once: fs::absolute asks the kernel for it on every call (20 000 calls for 10 000 files)
        auto key = [&](const fs::path &q) { return (q.is_absolute() || cwd_ec ? q : cwd / q).lexically_normal().string(); };
        std::unordered_set<std::string> inputs;
        for (auto &f : files) inputs.insert(key(f));
        for (size_t i = 0; i < files.size() && read_ahead; i++) {
            fs::path outdir_in = o.same_folder_as_input ? files[i].parent_path() : *o.output, dir;
            std::string name;
            const bool same = o.same_folder_as_input || outdir_in == *base;
            if (!compute_output_full_path(outdir_in, files[i], *base, o.keep_structure, suffix, o.format, same, dir, name)) continue;
            const std::string out = key(dir / name);
            if (out != key(files[i]) && inputs.count(out)) read_ahead = false;
        }
    }
    const auto t_plan = now();
    // ---- stage 2 (device): the engine calls of compressor.rs:287-306, batched.  Files that share a parameter set form one batch per device; the batches of
    // all windows go through ONE queue that a few host threads per device drain (round 6; before, every window ended in a join with the device idle while its
    // last batches were written): a window's batches are queued as soon as it is read, while the window before it is still finishing.  What is held in
    // memory stays two windows: window N + 1 is read only when window N - 1 is done.
    // CSH_CLI_SAME_DEVICE=1 (tests): --gpus N deals the batches over N device slots that are all device 0 -- the code path of N GPUs on a box that has one
    const bool same_device = getenv("CSH_CLI_SAME_DEVICE") != nullptr;
    const int ndev = o.dry_run ? 1 : (same_device ? std::max(1, o.gpus) : std::max(1, std::min(o.gpus, std::max(1, csh_device_count()))));
    // host threads per device, each with its own batches: while one batch is in its kernels the others are being parsed and uploaded or fetched and
    // written (separate streams; the boundary call is thread-safe).  A cold process pays for every thread's pools once (~55 ms): three for a short run
    // (2048 x 1080p: 0.75-0.83 s with 2, 0.78-0.84 with 3, 0.81-0.85 s with 4), four for a long one (10 000 files: 1.86-2.05 s with 2, 1.64-1.67 s with 4)
    const size_t per_dev = getenv("CSH_CLI_WORKERS") ? std::max<size_t>(1, size_t(atol(getenv("CSH_CLI_WORKERS")))) : (files.size() >= 4096 ? 4 : 3);   // (three: a tree of JPEG, PNG and WebP files is three batches of different kinds, the PNG and WebP ones latency-bound -- configs[4]: 2.1 s with two workers)
    const size_t nworkers = size_t(ndev) * per_dev;
    struct QueuedBatch { size_t window; std::vector<size_t> idx; };
    std::mutex qmu;
    std::condition_variable qcv, donecv;
    std::deque<QueuedBatch> queue;
    std::vector<size_t> pending(windows.size(), 0);   // batches of a window still in the queue or in a worker's hands
    bool closing = false;
    auto run_batch = [&](size_t dev, const std::vector<size_t> &idx) {
        std::vector<CByteArray> in(idx.size()), out(idx.size());
        std::vector<CCSResult> res(idx.size());
        std::vector<char> silent(idx.size(), 0);
        for (size_t k = 0; k < idx.size(); k++) { in[k].data = jobs[idx[k]].data.data(); in[k].length = jobs[idx[k]].data.size(); }
        CCSParameters p = jobs[idx[0]].params;
        if (o.format != Format::Original && !o.max_size) {
            cs_batch_convert(in.data(), in.size(), &p, map_format(o.format), int(dev), out.data(), res.data());   // the whole group in one device batch
        } else if (o.format != Format::Original) {
            // convert + size targeting: one engine call per file, as the reference does
            for (size_t k = 0; k < idx.size(); k++) {
                // convert, then walk the quality over the converted file (a WebP made here is decoded again on the device, as the reference does with libwebp)
                res[k] = cs_convert_in_memory(in[k].data, in[k].length, &p, map_format(o.format), &out[k]);
                if (o.max_size && res[k].success) {
                    CByteArray conv = out[k];
                    CCSParameters pk = p;
                    cs_free_result(&res[k]);
                    res[k] = cs_compress_to_size_in_memory(conv.data, conv.length, &pk, *o.max_size, true, &out[k]);
                    cs_free_bytes(&conv);
                } else if (o.max_size) {   // `.ok()?` in the reference: the error text is dropped
                    cs_free_result(&res[k]);
                    res[k].success = false; res[k].error_message = nullptr; silent[k] = 1;
                }
            }
        } else if (o.max_size) cs_batch_compress_to_size(in.data(), in.size(), &p, *o.max_size, true, int(dev), out.data(), res.data());
        else cs_batch_compress(in.data(), in.size(), &p, int(dev), out.data(), res.data());
        for (size_t k = 0; k < idx.size(); k++) {
            Job &j = jobs[idx[k]];
            j.ok = res[k].success;
            if (j.ok) j.result = out[k];
            else {
                if (!silent[k]) j.engine_msg = std::string("Error compressing file: ") + (res[k].error_message ? res[k].error_message : "");
                cs_free_bytes(&out[k]);
            }
            cs_free_result(&res[k]);
        }
        const auto t_w = now();
        // (the batch's inputs go here, on the worker: a window's 1024 buffers released in one go at its end were 25 ms of munmap with the device idle)
        parallel_for(idx.size(), std::min<size_t>(threads, 8), [&](size_t k) { std::vector<uint8_t>().swap(jobs[idx[k]].data); finish_job(idx[k]); });
        us_write += (long long)(ms(t_w, now()) * 1000.0);
    };
    std::vector<std::thread> workers;
    if (!o.dry_run) for (size_t w = 0; w < nworkers; w++) workers.emplace_back([&, w] {
        const size_t dev = same_device ? 0 : w % size_t(ndev);
        for (;;) {
            QueuedBatch qb;
            {
                std::unique_lock<std::mutex> lk(qmu);
                qcv.wait(lk, [&] { return closing || !queue.empty(); });
                if (queue.empty()) return;
                qb = std::move(queue.front()); queue.pop_front();
            }
            run_batch(dev, qb.idx);
            { std::lock_guard<std::mutex> lk(qmu); pending[qb.window]--; }
            donecv.notify_all();
        }
    });
    auto wait_window = [&](size_t wi) { std::unique_lock<std::mutex> lk(qmu); donecv.wait(lk, [&] { return pending[wi] == 0; }); };
    const auto t_engine0 = now();
    std::future<void> ahead;
    if (read_ahead) ahead = std::async(std::launch::async, read_window, windows[0].first, windows[0].second);
    for (size_t wi = 0; wi < windows.size(); wi++) {
        const size_t w0 = windows[wi].first, w1 = windows[wi].second;
        if (read_ahead) ahead.get(); else read_window(w0, w1);
        if (!o.dry_run) {
            // resize target (the only per-file parameter) and file type -> job indices.  The type is part of the key since round 6: a PNG or WebP batch pays a
            // latency that does not depend on its size (one workgroup per zlib stream / per VP8 frame: 0.3 / 0.8 s), so a tree's PNG and WebP files go to
            // the device in as few batches as the window allows instead of being cut wherever 128 files of the sorted tree end (configs[4]'s 96 + 96 + 96
            // files were three mixed batches, the WebP latency paid twice)
            auto kind_of = [](const std::vector<uint8_t> &d) -> uint32_t {
                if (d.size() >= 12 && !memcmp(d.data(), "RIFF", 4) && !memcmp(d.data() + 8, "WEBP", 4)) return 2;
                if (d.size() >= 8 && !memcmp(d.data(), "\x89PNG", 4)) return 1;
                return 0;
            };
            std::map<std::tuple<uint32_t, uint32_t, uint32_t>, std::vector<size_t>> groups;
            for (size_t i = w0; i < w1; i++) if (jobs[i].engine) groups[{kind_of(jobs[i].data), jobs[i].params.width, jobs[i].params.height}].push_back(i);
            std::vector<QueuedBatch> batches;
            // files per device batch (and never more than one device batch takes by bytes / declared pixels: cs_batch_extent).  A cold process pays for the
            // device pools it allocates (~25 MB per 1080p file) before the first kernel runs, and later batches reuse the first ones' pools: smaller
            // batches start sooner -- 2048 x 1080p files end to end on one MI355X: 3.3-5.6 s at 1024 files per batch, 1.0 s at 256, 0.9 s at 128; CSH_CLI_BATCH overrides
            // (a WebP output walks every picture macroblock step by step -- libwebp's encoder, one workgroup per picture: ~0.1 s for a 1500 x 844 picture however few
            // pictures share the device; its batches are as large as the window allows)
            const size_t kBatch = getenv("CSH_CLI_BATCH") ? std::max<size_t>(1, size_t(atol(getenv("CSH_CLI_BATCH")))) : (o.format == Format::Webp ? 1024 : 128);   // (round 5: 10 000 files 2.11-2.13 s at 128, 2.23-2.37 s at 256 -- half the pools a cold process has to map before its first kernel)
            for (auto &g : groups) {
                const size_t kGroupBatch = (std::get<0>(g.first) != 0 && !getenv("CSH_CLI_BATCH")) ? 1024 : kBatch;   // PNG / WebP inputs: latency-bound rows, the library cuts by memory itself
                std::vector<CByteArray> gin(g.second.size());
                for (size_t k = 0; k < gin.size(); k++) { gin[k].data = jobs[g.second[k]].data.data(); gin[k].length = jobs[g.second[k]].data.size(); }
                for (size_t k = 0, n = 0; k < g.second.size(); k += n) {
                    n = std::max<size_t>(1, cs_batch_extent(gin.data() + k, std::min(kGroupBatch, g.second.size() - k)));
                    batches.push_back(QueuedBatch{wi, std::vector<size_t>(g.second.begin() + k, g.second.begin() + k + n)});
                }
            }
            {
                std::lock_guard<std::mutex> lk(qmu);
                pending[wi] = batches.size();
                for (auto &qb : batches) queue.push_back(std::move(qb));
            }
            qcv.notify_all();
            // two windows in memory: the one just queued and the one before it (still finishing).  The window before must be done before the next is read;
            // without read-ahead (an output of one file may be another's input) the windows run strictly one after the other
            if (!read_ahead) wait_window(wi);
            else if (wi > 0) wait_window(wi - 1);
        }
        if (read_ahead && wi + 1 < windows.size()) ahead = std::async(std::launch::async, read_window, windows[wi + 1].first, windows[wi + 1].second);
    }
    if (!o.dry_run) {
        for (size_t wi = 0; wi < windows.size(); wi++) wait_window(wi);
        { std::lock_guard<std::mutex> lk(qmu); closing = true; }
        qcv.notify_all();
        for (auto &t : workers) t.join();
    }
    ms_engine = ms(t_engine0, now()); ms_write = double(us_write.load()) / 1000.0;
    ms_read = double(us_read.load()) / 1000.0;
    const auto t_done = now();
    if (trace) fprintf(stderr, "[cli] %zu files: scan %.0f ms, read+prepare %.0f ms (summed over the windows; all but the first behind the engine), first read + engine + write %.0f ms (of which policy+write, summed over the batch workers: %.0f ms)\n", files.size(), ms(t_start, t_scan),
                       ms_read, ms_engine, ms_write);
    if (o.json) printf("%s\n", build_json(results, o.dry_run, nullptr).c_str());
    else fputs(build_recap(results, verbose, isatty(1)).c_str(), stdout);
    if (trace) fprintf(stderr, "[cli] run() took %.0f ms (windows planned at %.0f, last window done at %.0f)\n", ms(t_start, now()), ms(t_start, t_plan), ms(t_start, t_done));
    return 0;
}

}  // namespace cli
