// bioem_hip_plugin.cpp -- TEST INFRASTRUCTURE ONLY: the proof that libbioem_hip.so drops into the reference.
//
// This is the `class bioem_hip : public bioem` of INTEGRATION.md, compiled.  `make -C oracle ref_hip` builds
// oracle/_ref/bioEM_ref_hip from the UNMODIFIED reference sources where they lie under /root/reference
// (-DWITH_CUDA, no bioem_cuda.cu) plus this one file: main.cpp:80-89 then calls bioem_cuda_create()
// (include/bioem_cuda.h:20) when GPU=1, receives the subclass below, and the reference's own run() loop
// (bioem.cpp:763-891: createProjection / createConvolutedProjectionMap on the host, compareRefMaps through the
// virtual at bioem.cpp:853) drives the MI355X engine through the C ABI of include/bioem_hip.h.
// The product links nothing from here.
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "bioem.h"      // /root/reference/include
#include "bioem_cuda.h" // declares bioem *bioem_cuda_create()
#include "bioem_hip.h"  // this repository: include/bioem_hip.h

extern int mpi_rank; // main.cpp

static_assert(sizeof(mycomplex_t) == 2 * sizeof(float), "mycomplex_t must be float[2]");
static_assert(sizeof(myparam5_t) == sizeof(bioem_hip_param5), "myparam5_t layout (defs.h:128-135)");
static_assert(sizeof(bioem_Probability_map) == sizeof(bioem_hip_prob_map), "bioem_Probability_map layout (map.h:116-128)");
static_assert(sizeof(bioem_Probability_angle) == sizeof(bioem_hip_prob_angle), "bioem_Probability_angle layout (map.h:130-135)");
static_assert(offsetof(bioem_param_device, tousepsf) == offsetof(bioem_hip_param_device, tousepsf),
              "bioem_param_device layout (param.h:26-47)");

class bioem_hip : public bioem
{
public:
  bioem_hip() : h(NULL) {}
  ~bioem_hip()
  {
    if (h)
      bioem_hip_destroy(h); // == bioem_cuda::deviceExit (bioem_cuda.cu:1023-1053)
  }

  // == bioem_cuda::compareRefMaps (bioem_cuda.cu:527-684); same arguments, same 2-slot buffer convention
  int compareRefMaps(int iPipeline, int iOrient, int iConv, int maxParallelConv, mycomplex_t *conv_mapsFFT,
                     myparam5_t *comp_params, const int startMap = 0)
  {
    if (startMap)
      myError("startMap not implemented for GPU code");
    chk(bioem_hip_compare(h, iPipeline, iOrient, iConv, maxParallelConv, param.nTotParallelConv,
                          (const float *) conv_mapsFFT, (const bioem_hip_param5 *) comp_params));
    return 0;
  }
  void *malloc_device_host(size_t size) { return bioem_hip_host_alloc(size); } // bioem.h:56
  void free_device_host(void *ptr) { bioem_hip_host_free(ptr); }               // bioem.h:57
  void rebalance(int) {}                                                       // 100 % of the particles stay on the device

protected:
  int deviceInit() // bioem_cuda.cu:818-951
  {
    bioem_hip_param_device pd;
    memset(&pd, 0, sizeof(pd));
    memcpy(&pd, &param.param_device, offsetof(bioem_hip_param_device, tousepsf));
    pd.tousepsf = param.param_device.tousepsf ? 1 : 0;
    const int ndev = bioem_hip_device_count();
    const int dev = (getenv("GPUDEVICE") && atoi(getenv("GPUDEVICE")) >= 0) ? atoi(getenv("GPUDEVICE")) : mpi_rank % ndev;
    chk(bioem_hip_create(&h, dev, &pd, RefMap.ntotRefMap, param.nTotGridAngles, param.nTotCTFs, BioEMAlgo));
    chk(bioem_hip_upload_particles(h, (const float *) RefMap.RefMapsFFT, RefMap.sum_RefMap, RefMap.sumsquare_RefMap));
    printf("BioEM HIP plugin: device %d, comparison kernel %s\n", dev, bioem_hip_kernel_name(h));
    return 0;
  }
  int deviceStartRun() // bioem_cuda.cu:953-1011
  {
    chk(bioem_hip_start_run(h, pProb.ptr));
    return 0;
  }
  int deviceFinishRun() // bioem_cuda.cu:1013-1021
  {
    chk(bioem_hip_finish_run(h, pProb.ptr));
    return 0;
  }

private:
  void chk(int rc)
  {
    if (rc)
      myError("%s", bioem_hip_last_error(h));
  }
  bioem_hip_handle h;
};

// the factory main.cpp:83 calls under -DWITH_CUDA (reference: bioem_cuda.cu:1073-1086)
bioem *bioem_cuda_create()
{
  if (bioem_hip_device_count() == 0)
  {
    printf("No HIP device available, using fallback to CPU version\n");
    return new bioem;
  }
  return new bioem_hip;
}
