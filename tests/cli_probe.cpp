// cli_probe.cpp -- test-only front end to the functions of caesium-clt_amd/cli/cli.hpp, one call per process:
//   cli_probe bytesize <text>                 -> "<bytes>" | "ERR"
//   cli_probe fmtsize <bytes>                 -> display string
//   cli_probe minsavings <text>               -> "pct <v>" | "bytes <n>" | "ERR <message>"
//   cli_probe base <path>...                  -> base folder after folding the paths ("NONE" if unset)
//   cli_probe outpath <outdir> <input> <base> <keep 0|1> <suffix> <format> <same 0|1>  -> "<dir>\n<name>" | "NONE"
//   cli_probe dims <file> <keep_metadata 0|1> -> "<w> <h>" | "ERR"
//   cli_probe threads <requested> <available> -> count
//   cli_probe args <flags...>                 -> "OK" + a dump of the parsed options | "ERR <message>"
//   cli_probe json <dry 0|1> [<orig> <out> <osize> <csize> <status 0..2> <msg>]...
//   cli_probe recap <verbose> [<same 6-tuples>]...
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>

#include "cli.hpp"

using namespace cli;

static Format fmt_of(const std::string &s) {
    if (s == "jpeg") return Format::Jpeg;
    if (s == "png") return Format::Png;
    if (s == "gif") return Format::Gif;
    if (s == "webp") return Format::Webp;
    if (s == "tiff") return Format::Tiff;
    return Format::Original;
}
static std::vector<Result> results_from(char **a, int n) {
    std::vector<Result> r;
    for (int i = 0; i + 6 <= n; i += 6) {
        Result x; x.original_path = a[i]; x.output_path = a[i + 1]; x.original_size = strtoull(a[i + 2], nullptr, 10); x.compressed_size = strtoull(a[i + 3], nullptr, 10);
        x.status = Status(atoi(a[i + 4])); x.message = a[i + 5];
        r.push_back(x);
    }
    return r;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    std::string c = argv[1];
    if (c == "bytesize") { uint64_t v; if (parse_bytesize(argv[2], v)) printf("%llu\n", (unsigned long long)v); else puts("ERR"); }
    else if (c == "fmtsize") puts(format_bytesize(strtoull(argv[2], nullptr, 10)).c_str());
    else if (c == "minsavings") { MinSavings m; std::string e; if (!parse_min_savings(argv[2], m, e)) printf("ERR %s\n", e.c_str()); else if (m.percent) printf("pct %.6f\n", m.pct); else printf("bytes %llu\n", (unsigned long long)m.bytes); }
    else if (c == "base") { std::optional<fs::path> b; for (int i = 2; i < argc; i++) { auto nb = compute_base_folder(b, argv[i]); if (nb) b = nb; } puts(b ? b->string().c_str() : "NONE"); }
    else if (c == "outpath") {
        fs::path d; std::string n;
        if (compute_output_full_path(argv[2], argv[3], argv[4], atoi(argv[5]), argv[6], fmt_of(argv[7]), atoi(argv[8]), d, n)) printf("%s\n%s\n", d.string().c_str(), n.c_str()); else puts("NONE");
    }
    else if (c == "dims") {
        std::ifstream f(argv[2], std::ios::binary); std::vector<uint8_t> b((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        size_t w, h; if (probe_dimensions(b, atoi(argv[3]), w, h)) printf("%zu %zu\n", w, h); else puts("ERR");
    }
    else if (c == "threads") printf("%zu\n", parallelism_count(uint32_t(atoi(argv[2])), size_t(atoi(argv[3]))));
    else if (c == "args") {
        Options o; std::string e; std::vector<std::string> a(argv + 2, argv + argc);
        if (!parse_args(a, o, e)) { printf("ERR %s\n", e.c_str()); return 0; }
        printf("OK quality=%d lossless=%d max_size=%lld width=%d height=%d long=%d short=%d no_upscale=%d output=%s same=%d format=%d png=%d chroma=%d baseline=%d zopfli=%d exif=%d "
               "keep_dates=%d strip_icc=%d suffix=%s recursive=%d keep_structure=%d dry=%d threads=%u extonly=%d overwrite=%d minsav=%s quiet=%d verbose=%d json=%d gpus=%d files=%zu\n",
               o.quality ? int(*o.quality) : -1, o.lossless, o.max_size ? (long long)*o.max_size : -1ll, o.width ? int(*o.width) : -1, o.height ? int(*o.height) : -1,
               o.long_edge ? int(*o.long_edge) : -1, o.short_edge ? int(*o.short_edge) : -1, o.no_upscale, o.output ? o.output->string().c_str() : "-", o.same_folder_as_input, int(o.format),
               o.png_opt_level, o.chroma, o.jpeg_baseline, o.zopfli, o.exif, o.keep_dates, o.strip_icc, o.suffix ? o.suffix->c_str() : "-", o.recursive, o.keep_structure, o.dry_run, o.threads,
               o.check_extension_only, int(o.overwrite), o.min_savings ? (o.min_savings->percent ? "pct" : "bytes") : "-", o.quiet, o.verbose, o.json, o.gpus, o.files.size());
    }
    else if (c == "json") puts(build_json(results_from(argv + 3, argc - 3), atoi(argv[2]), nullptr).c_str());
    else if (c == "recap") fputs(build_recap(results_from(argv + 3, argc - 3), atoi(argv[2]), false).c_str(), stdout);
    else return 2;
    return 0;
}
