// main.cpp -- `bioEM` command line of the MI355X engine; same options and phases as the reference's
// main (/root/reference/main.cpp:57-134): configure, run, report the wall time.
#include <chrono>
#include <cstdio>

#include "bioem_host.h"

int main(int argc, char **argv)
{
  bioem_host::Driver bio;
  printf("Configuring\n");
  if (bio.configure(argc, argv) == 0)
  {
    printf("Running\n");
    const auto t0 = std::chrono::steady_clock::now();
    bio.run();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("The code ran for %f seconds (rank %d).\n", s, 0);
    bio.cleanup();
  }
  return 0;
}
