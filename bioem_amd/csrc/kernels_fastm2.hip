// kernels_fastm2.hip -- the k_compare_fastm2 instantiations of kernel_table.inc (33..47-row windows)
#include "engine_types.hpp"
#include "posterior.hpp"
#include "fft_registers.hpp"
#include "compare_args.hpp"
#include "compare_fast.hpp"
#include "compare_fastm.hpp"
#include "compare_fastm2.hpp"
#define K_FASTM2(R, NYQ, GS) {KF_FASTM2, {R, NYQ, GS, 0, 0, 0}, reinterpret_cast<const void *>(k_compare_fastm2<R, NYQ, GS>)},
#define BIOEM_FAMILY_FN bioem_kernels_fastm2
#include "kernels_family.inc"
