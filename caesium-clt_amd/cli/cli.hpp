// cli.hpp -- the caesiumclt command-line shell (SURVEY.md 8f "next" rank 1), host-side C++17.
// Mirrors, flag for flag and message for message, the reference's CLI layer:
//   flags + validators        /root/reference/src/options.rs:47-257
//   input scan + base path    /root/reference/src/scan_files.rs:8-143
//   per-file policy           /root/reference/src/compressor.rs:103-184, 190-257, 317-409, 448-561
//   recap + JSON              /root/reference/src/main.rs:15-285
// The engine calls (compressor.rs:287-306) go to libcaesium_hip's C ABI in device batches instead of one call per
// rayon worker.
#pragma once
#include <cstdint>
#include <filesystem>
#include <optional>
#include <string>
#include <vector>

namespace cli {
namespace fs = std::filesystem;

enum class Overwrite { All, Never, Bigger };
enum class Format { Jpeg, Png, Gif, Webp, Tiff, Original };
struct MinSavings { bool percent = false; double pct = 0; uint64_t bytes = 0; };

struct Options {
    std::optional<uint32_t> quality;
    bool lossless = false;
    std::optional<size_t> max_size;
    std::optional<uint32_t> width, height, long_edge, short_edge;
    bool no_upscale = false;
    std::optional<fs::path> output;
    bool same_folder_as_input = false;
    Format format = Format::Original;
    int png_opt_level = 3;
    int chroma = 0;  // 444, 422, 420, 411, 0 = auto
    bool jpeg_baseline = false, zopfli = false, exif = false, keep_dates = false, strip_icc = false;
    std::optional<std::string> suffix;
    bool recursive = false, keep_structure = false, dry_run = false;
    uint32_t threads = 0;
    bool check_extension_only = false;
    Overwrite overwrite = Overwrite::All;
    std::optional<MinSavings> min_savings;
    bool quiet = false;
    int verbose = 1;
    bool json = false;
    int gpus = 1;  // extension: devices to shard the batch over
    std::vector<std::string> files;
    bool help = false, version = false;
};

// clap-equivalent parsing; on failure returns false with a message (caller prints it and exits with 2)
bool parse_args(const std::vector<std::string> &argv, Options &o, std::string &err);
std::string usage();

bool parse_bytesize(const std::string &s, uint64_t &out);          // bytesize 2.x FromStr: "100KB", "0.5MiB", "123"
bool parse_min_savings(const std::string &s, MinSavings &out, std::string &err);
std::string format_bytesize(uint64_t n);                            // bytesize 2.x Display: "293.9 KiB"

// scan_files.rs
bool has_supported_extension(const fs::path &p);
bool is_filetype_supported(const fs::path &p);
std::optional<fs::path> compute_base_folder(const std::optional<fs::path> &bf, const fs::path &new_path);
void scan_files(const std::vector<std::string> &args, bool recursive, bool check_extension_only, std::optional<fs::path> &base,
                std::vector<fs::path> &files);

// compressor.rs:448-501
bool compute_output_full_path(const fs::path &output_directory, const fs::path &input_file, const fs::path &base_directory,
                              bool keep_structure, const std::string &suffix, Format format, bool same_folder_as_input,
                              fs::path &dir_out, std::string &name_out);

// imagesize::blob_size + EXIF orientation (compressor.rs:538-561)
bool probe_dimensions(const std::vector<uint8_t> &buf, bool keep_metadata, size_t &w, size_t &h);

enum class Status { Success, Skipped, Error };
struct Result {
    std::string original_path, output_path;
    uint64_t original_size = 0, compressed_size = 0;
    Status status = Status::Error;
    std::string message;
};
std::string build_json(const std::vector<Result> &results, bool dry_run, const char *error);
std::string build_recap(const std::vector<Result> &results, int verbose, bool color);
size_t parallelism_count(uint32_t requested, size_t available);

int run(const Options &o);  // the whole program after parsing; returns the process exit code
}  // namespace cli
