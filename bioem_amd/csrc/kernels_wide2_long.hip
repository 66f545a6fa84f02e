// kernels_wide2_long.hip -- the k_compare_wide2 instantiations of kernel_table.inc with a register FFT of length (R) >= 20
#include "engine_types.hpp"
#include "posterior.hpp"
#include "fft_registers.hpp"
#include "compare_args.hpp"
#include "compare_fast.hpp"
#include "compare_wide2.hpp"
namespace
{
// only the lengths of this translation unit are instantiated; the other lines of the table give a null stub
template <int R, int NRW, int NBLK, bool NYQ, int HALVES, int NW, bool MINE>
struct Wide2Stub
{
  static const void *fn() { return nullptr; }
};
template <int R, int NRW, int NBLK, bool NYQ, int HALVES, int NW>
struct Wide2Stub<R, NRW, NBLK, NYQ, HALVES, NW, true>
{
  static const void *fn() { return reinterpret_cast<const void *>(k_compare_wide2<R, NRW, NBLK, NYQ, HALVES, NW>); }
};
} // namespace
#define K_WIDE2(R, NRW, NBLK, NYQ, HALVES, NW)                                                                     \
  {KF_WIDE2, {R, NRW, NBLK, NYQ, HALVES, NW}, Wide2Stub<R, NRW, NBLK, NYQ, HALVES, NW, ((R) >= 20)>::fn()},
#define BIOEM_FAMILY_FN bioem_kernels_wide2_long
#include "kernels_family.inc"
