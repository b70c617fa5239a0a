// fasta.h -- host FASTA/FASTQ(.gz) reader (kseq semantics), see fasta.cpp.
#pragma once
#include <cstdint>
#include <memory>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/tracs_hip.h"

namespace tracs {

// std::vector that does not zero on resize(): the parallel readers size n * L bytes at once and every byte is written by
// the worker that owns its record, so the pages are first touched in parallel instead of memset by one thread
template <class T>
struct default_init_allocator : std::allocator<T> {
    template <class U> struct rebind { using other = default_init_allocator<U>; };
    template <class U, class... A> void construct(U *p, A &&...a)
    {
        if constexpr (sizeof...(A) == 0) ::new (static_cast<void *>(p)) U;
        else ::new (static_cast<void *>(p)) U(std::forward<A>(a)...);
    }
};
using ByteVec = std::vector<uint8_t, default_init_allocator<uint8_t>>;

struct FastaData {
    size_t n = 0, L = 0;
    ByteVec seq;                       // n * L raw bytes (case preserved; the pack kernel folds case)
    std::vector<std::string> names;
};

// Appends the records of `path` to `out` (so two files can share one FastaData).  Returns
// TRACS_OK or TRACS_E_OPEN / TRACS_E_FASTA / TRACS_E_RAGGED with `err` set to the message.
int read_fasta(const std::string &path, FastaData &out, std::string &err);

}  // namespace tracs
