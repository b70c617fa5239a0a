// fasta.h -- host FASTA/FASTQ(.gz) reader (kseq semantics), see fasta.cpp.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/tracs_hip.h"

namespace tracs {

struct FastaData {
    size_t n = 0, L = 0;
    std::vector<uint8_t> seq;          // n * L raw bytes (case preserved; the pack kernel folds case)
    std::vector<std::string> names;
};

// Appends the records of `path` to `out` (so two files can share one FastaData).  Returns
// TRACS_OK or TRACS_E_OPEN / TRACS_E_FASTA / TRACS_E_RAGGED with `err` set to the message.
int read_fasta(const std::string &path, FastaData &out, std::string &err);

}  // namespace tracs
