// kernels_fast.hip -- the k_compare_fast instantiations of kernel_table.inc (windows of at most 21 rows)
#include "engine_types.hpp"
#include "posterior.hpp"
#include "fft_registers.hpp"
#include "compare_args.hpp"
#include "compare_fast.hpp"
#define K_FAST(WD, R, NYQ, GS) {KF_FAST, {WD, R, NYQ, GS, 0, 0}, reinterpret_cast<const void *>(k_compare_fast<WD, R, NYQ, GS>)},
#define BIOEM_FAMILY_FN bioem_kernels_fast
#include "kernels_family.inc"
