// main.cpp -- `caesiumclt` entry point (reference: /root/reference/src/main.rs:46-111).
#include <cstdio>

#include "cli.hpp"

int main(int argc, char **argv) {
    std::vector<std::string> args(argv + 1, argv + argc);
    cli::Options o;
    std::string err;
    if (!cli::parse_args(args, o, err)) {
        fprintf(stderr, "error: %s\n\n%s\nFor more information, try '--help'.\n", err.c_str(), "Usage: caesiumclt [OPTIONS] <--quality <QUALITY>|--lossless|--max-size <MAX_SIZE>> <--output <OUTPUT>|--same-folder-as-input> [FILES]...");
        return 2;
    }
    if (o.help) { fputs(cli::usage().c_str(), stdout); return 0; }
    if (o.version) { puts("caesiumclt 1.3.0 (caesium-hip, gfx950)"); return 0; }
    return cli::run(o);
}
