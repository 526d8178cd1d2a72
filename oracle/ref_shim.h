// Force-included when the REFERENCE sources are compiled for the CPU (oracle/Makefile, target ref).
// The reference's own HAS_NO_CUDA mode (tfqmrgpu.hxx:28-32 + its tfqmrgpu_cudaStubs.hxx) is what is
// being built; nothing of CUDA is emulated here.  This header only repairs three things that keep
// that mode from compiling / running with g++ 11:
//   1. standard headers that nvcc would have pulled in implicitly;
//   2. two identifiers used by tfqmrgpu.cu:573,581 that the reference's stub header forgot to stub
//      (a no-op on the CPU: it sets a CUDA shared-memory attribute);
//   3. the reference's bump allocator aligns ABSOLUTE addresses to 256 B (tfqmrgpu_util.hxx:56-63)
//      while its stubbed cudaMalloc is plain malloc (16 B aligned), so windows computed from address 0
//      do not match -> hand out 256-B aligned memory.
#include <cstdint>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cassert>
#include <cmath>
#include <vector>
#include <string>
#include <iostream>
#include <fstream>
#include <sstream>
#include <algorithm>
#include <complex>
#include <numeric>
#include <limits>
#include <omp.h>
#define cudaFuncSetAttribute(f, a, v) ((void)0)
#define cudaFuncAttributeMaxDynamicSharedMemorySize 0
#define malloc(n) aligned_alloc(256, ((size_t(n) + 255) / 256) * 256)
