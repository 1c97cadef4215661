// Host build of the PRODUCT's a-contrario arithmetic (coloc_amd/csrc/clc_acr.h: the portable log10, the per-iteration sampler in its
// run-time and compile-time-size forms, the NFA term) as a tiny shared library for the tests.  Since round 6 the oracle
// (oracle/clc_oracle_acr.c) no longer includes that header: tests/test_acransac.py holds the two statements against each other
// through this library.  Test infrastructure only.
#include "../../coloc_amd/csrc/clc_acr.h"

extern "C" double acr_host_log10(double x) { return clc_acr_log10(x); }
extern "C" void acr_host_sample(uint64_t seed, uint32_t iter, uint32_t n_index, int m, uint32_t* pos) { clc_acr_sample(seed, iter, n_index, m, pos); }
extern "C" void acr_host_sample_fixed(uint64_t seed, uint32_t iter, uint32_t n_index, int m, uint32_t* pos)
{
    if (m == 3) { uint32_t p[3]; clc_acr_sample_t<3>(seed, iter, n_index, p); for (int j = 0; j < 3; ++j) pos[j] = p[j]; }
    else { uint32_t p[5]; clc_acr_sample_t<5>(seed, iter, n_index, p); for (int j = 0; j < 5; ++j) pos[j] = p[j]; }
}
extern "C" double acr_host_nfa(double loge0, double logalpha0, double mult, double e_k, int k, int m, float logc_n_k, float logc_k_k)
{
    return clc_acr_nfa(loge0, logalpha0, mult, e_k, k, m, logc_n_k, logc_k_k);
}
