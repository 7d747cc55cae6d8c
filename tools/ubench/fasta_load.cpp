// fasta_load.cpp -- ipcr_genome_add_fasta timed from a native process (the system's HIP runtime, no interpreter):
//   fasta_load <file.fa> [repeats]
// (development tool: tells a loader limit from a limit of the process it runs in; build: tools/ubench/build.sh)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <sys/stat.h>

#include "ipcr_hip.h"

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    struct stat sb;
    if (stat(argv[1], &sb)) return 2;
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    for (int r = 0; r < reps; ++r) {
        ipcr_genome *g = nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        if (ipcr_genome_create((uint64_t)sb.st_size + (1u << 20), 4096, &g) != IPCR_OK) { fprintf(stderr, "%s\n", ipcr_last_error()); return 1; }
        uint32_t n = 0;
        size_t need = 0;
        if (ipcr_genome_add_fasta(g, argv[1], &n, nullptr, 0, &need) != IPCR_OK) { fprintf(stderr, "%s\n", ipcr_last_error()); return 1; }
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("native load %d: %u records, %llu bases, %.2f ms, %.2f GB/s of file, %.2f Gbases/s\n", r, n,
               (unsigned long long)ipcr_genome_total_bases(g), s * 1e3, sb.st_size / s / 1e9, ipcr_genome_total_bases(g) / s / 1e9);
        ipcr_genome_destroy(g);
    }
    return 0;
}
