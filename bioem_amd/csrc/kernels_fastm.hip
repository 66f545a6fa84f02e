// kernels_fastm.hip -- the k_compare_fastm instantiations of kernel_table.inc (27- / 31-row windows, matrix-core window pass)
#include "engine_types.hpp"
#include "posterior.hpp"
#include "fft_registers.hpp"
#include "compare_args.hpp"
#include "compare_fast.hpp"
#include "compare_fastm.hpp"
#define K_FASTM(WD, R, NYQ, GS) {KF_FASTM, {WD, R, NYQ, GS, 0, 0}, reinterpret_cast<const void *>(k_compare_fastm<WD, R, NYQ, GS>)},
#define BIOEM_FAMILY_FN bioem_kernels_fastm
#include "kernels_family.inc"
