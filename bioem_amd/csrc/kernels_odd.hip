// kernels_odd.hip -- odd image sizes: the k_compare_rows / k_compare_oddfft instantiations of kernel_table.inc
#include "engine_types.hpp"
#include "posterior.hpp"
#include "fft_registers.hpp"
#include "compare_args.hpp"
#include "compare_fast.hpp"
#include "compare_rows.hpp"
#define K_ROWS(WD, GS) {KF_ROWS, {WD, GS, 0, 0, 0, 0}, reinterpret_cast<const void *>(k_compare_rows<WD, GS>)},
#define K_ODDFFT(WD, R) {KF_ODDFFT, {WD, R, 0, 0, 0, 0}, reinterpret_cast<const void *>(k_compare_oddfft<WD, R>)},
#define BIOEM_FAMILY_FN bioem_kernels_odd
#include "kernels_family.inc"
