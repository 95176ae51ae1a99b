// main.cpp -- `caesiumclt` entry point (reference: /root/reference/src/main.rs:46-111).
#include <cstdio>
#include <cstdlib>
#include <unistd.h>

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
    const int rc = cli::run(o);
    // Everything is written and reported.  The runtime's own exit handlers unmap the device pools block by block and tear the context down -- several
    // hundred milliseconds for a process that held a few GiB -- which the kernel does for a process that simply ends.  (CSH_CLI_SLOW_EXIT=1: the long way.)
    fflush(stdout); fflush(stderr);
    if (!getenv("CSH_CLI_SLOW_EXIT")) _exit(rc);
    return rc;
}
