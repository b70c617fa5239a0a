// rowwriter.h -- the CSV rows of `tracs distance` (tracs/distance.py:206-258), formatted and written by the library's host code
// (csrc/alignio.cpp).
#pragma once
#include <cstddef>
#include <cstdint>

namespace tracs {

class DistanceRowWriter {
public:
    DistanceRowWriter();
    ~DistanceRowWriter();
    DistanceRowWriter(const DistanceRowWriter &) = delete;
    DistanceRowWriter &operator=(const DistanceRowWriter &) = delete;
    // names[k]: sample k (must outlive the writer); rows are appended to `path`
    int open(const char *path, const char *const *names, size_t n_names, const char *ref);
    template <class T, class DeltaOf>
    int append(const T *rows, const T *cols, const T *snpd, const T *filt, const T *ncomp, DeltaOf delta_of, const double *p_direct,
               const double *e_k, size_t n, int with_dates, double k_max);
    // the columns as they leave the device; the date difference of a row from the samples' day numbers
    int append_u32(const uint32_t *rows, const uint32_t *cols, const uint32_t *snpd, const uint32_t *filt, const uint32_t *ncomp,
                   const int32_t *days, const double *p_direct, const double *e_k, size_t n, int with_dates, double k_max);
    int close();
    uint64_t written() const;

private:
    struct Impl;
    Impl *p;
};

}  // namespace tracs
