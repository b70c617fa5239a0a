// ref_shim.cpp -- builds the parts of the REFERENCE that compile in this image, from the
// reference's own headers where they lie (-I/root/reference/src), into oracle/_ref/.
//
// TEST INFRASTRUCTURE ONLY (see oracle/tracs_oracle.c header).  No reference source is
// copied: this file only #includes the reference headers and gives them a Python face.
//
//   src/transcluster.hpp   -- needs only libstdc++/libm (it relies on pybind11/stl.h having
//                             pulled in <vector>/<unordered_map>, as python_bindings.cpp does)
//   src/dmultinomial.hpp   -- needs pybind11 + numpy (both present in the image)
//   src/pairsnp.hpp        -- NOT built: needs boost/dynamic_bitset.hpp and
//                             boost/math/distributions/binomial.hpp, absent from the image.
//
// Module name `_tracs_ref`; function names are prefixed ref_ so nothing can mistake it for
// the product's `TRACS` module.  Extra entry points (ref_lprob_k_given_N_2, ref_expected_k,
// ref_upper_bound_E) expose the reference's internal functions for finer-grained pinning.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>
#include <pybind11/numpy.h>
#include <unordered_map>
#include <vector>
#include <tuple>
#include <cmath>

#include "transcluster.hpp"
#include "dmultinomial.hpp"

namespace py = pybind11;

static std::vector<double> ref_lgamma_table()
{
    std::vector<double> lg;   // same table trans_dist builds (transcluster.hpp:253-258)
    lg.reserve(10000);
    for (double i = 0; i < 10000; i++) lg.push_back(std::lgamma(i));
    return lg;
}

#ifndef REF_MODULE_NAME
#define REF_MODULE_NAME _tracs_ref
#endif
PYBIND11_MODULE(REF_MODULE_NAME, m)
{
    m.doc() = "reference TRACS transcluster/dmultinomial, compiled in place (oracle/_ref)";
    m.def("ref_trans_dist", [](const std::vector<int> &n, const std::vector<double> &d, double lamb,
                               double beta, double thr) { return trans_dist(n, d, lamb, beta, thr); });
    m.def("ref_lprob_k_given_N", [](size_t N, size_t k, double delta, double lamb, double beta,
                                    const std::vector<double> &lg) {
        return lprob_k_given_N(N, k, delta, lamb, beta, lg);
    });
    m.def("ref_lprob_k_given_N_2", [](size_t N, size_t k, double delta, double lamb, double beta) {
        return lprob_k_given_N_2(N, k, delta, lamb, beta, ref_lgamma_table());
    });
    m.def("ref_upper_bound_E", [](double delta, double lamb, double beta, size_t N) {
        return upper_bound_E(ref_lgamma_table(), delta, lamb, beta, N);
    });
    m.def("ref_expected_k", [](int N, double delta, double lamb, double beta, double thr) {
        std::unordered_map<std::tuple<int, int, double>, std::tuple<double, double>> memo;
        return expected_k(N, delta, lamb, beta, thr, ref_lgamma_table(), memo);
    });
    m.def("ref_calculate_posteriors", [](py::array_t<double> counts, std::vector<double> alphas, bool keep,
                                         double threshold) {
        return calculate_posteriors(counts, alphas, keep, threshold);
    });
}
