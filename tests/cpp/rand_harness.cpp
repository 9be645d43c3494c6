// compares bk::GlibcRand with the C library's rand() of a fresh process (prints the number of differing values)
#include <cstdio>
#include <cstdlib>
#include "../../biokanga_amd/csrc/host/glibc_rand.h"
int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 100000;
    bk::GlibcRand g;
    int bad = 0;
    for (int i = 0; i < n; i++) bad += g.next() != rand();
    printf("%d\n", bad);
    return bad != 0;
}
