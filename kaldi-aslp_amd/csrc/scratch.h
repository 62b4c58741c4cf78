// scratch.h -- grow-only device scratch slots (see runtime.cpp).
#pragma once
#include <stddef.h>
namespace aslp {
enum { kScratchReduce = 0, kScratchReduce2 = 1, kScratchGemm = 2, kScratchCtc = 3, kScratchMisc = 4, kNumScratch = 5 };
void *scratch(int slot, size_t bytes);
// zero-initialised, never-moving array of counters (column-reduce tickets); every user leaves its counters at 0
unsigned *tickets(int count);
}  // namespace aslp
