// test harness: SamPrealloc (biokanga_amd/csrc/host/cli_common.h) - the output file's pages made by background threads.
//   prealloc_harness <file> <estimate> <bytes written> <keep 0|1>
// starts the file, waits until the threads have ended (a hang here is the failure the test guards against), writes `bytes written`
// bytes of a pattern through the mapping (or with pwrite when there is none), cuts the file there when kept.
#include "../../biokanga_amd/csrc/host/cli_common.h"

int main(int argc, char **argv)
{
    if (argc != 5) return 2;
    const uint64_t est = strtoull(argv[2], nullptr, 10), put = strtoull(argv[3], nullptr, 10);
    const bool keep = atoi(argv[4]) != 0;
    bkcli::SamPrealloc pre;
    pre.start(argv[1], est);
    if (pre.fd < 0) return 3;
    for (int spins = 0; !pre.ended.load(); spins++) { if (spins > 60000) return 4; usleep(1000); }
    printf("done %lld mapped %d\n", (long long)pre.done.load(), pre.map ? 1 : 0);
    if ((uint64_t)pre.done.load() != est) return 5;
    std::vector<char> pat(put);
    for (uint64_t i = 0; i < put; i++) pat[i] = (char)('a' + i % 23);
    if (pre.map) memcpy(pre.map, pat.data(), put);
    else if (pwrite(pre.fd, pat.data(), put, 0) != (ssize_t)put) return 6;
    if (keep) { if (ftruncate(pre.fd, (off_t)put) != 0) return 7; pre.kept = true; }
    return 0;
}
