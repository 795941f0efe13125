// Stands where the reference's src/Benchmark.h stood: the Benchmark singleton
// (src/Benchmark.h:23-151) -- same methods, same table; the library's own stage brackets are
// routed into it from the first GetInstance() on.
#ifndef ARVX_DROPIN_BENCHMARK_H
#define ARVX_DROPIN_BENCHMARK_H
#include "arvx/benchmark.hpp"

using Benchmark = arvx::Benchmark;
#endif
