// fasta_hostpack.h -- see fasta_hostpack.cpp
#pragma once
#include <stdint.h>

#include <array>
#include <string>
#include <vector>

namespace ipcr {

struct FastaRecord {
    std::string id;
    uint64_t region = 0, bytes = 0; // the record's sequence text in the file
    uint32_t W = 0, lt = 1;         // bases per line, bytes per line end
    uint64_t full = 0, tail = 0;    // whole lines, bases of the last (shorter) line
    uint64_t len = 0;               // bases
};

struct FastaText {
    int fd = -1;
    const uint8_t *data = nullptr;
    size_t size = 0;
    std::vector<FastaRecord> records;
    ~FastaText();
    bool open(const char *path);
    bool pack(const FastaRecord &r, uint64_t *lo, uint64_t *hi, uint64_t *iv, uint64_t words, uint8_t *dirty, uint64_t group_cols) const;
};

} // namespace ipcr
