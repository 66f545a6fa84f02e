/* TEST INFRASTRUCTURE (oracle/): feeds a half-spectrum [N][N/2+1] (float2, arbitrary -- in particular NOT Hermitian
 * in columns 0 and N/2) to the image's FFTW3-API library (AMD hipFFTW; FFTW itself is absent from the image) through
 * the call the reference makes for its cross-correlation maps (fftwf_plan_dft_c2r_2d + fftwf_execute_dft_c2r,
 * /root/reference/bioem.cpp:1458, plan made at param.cpp:1521) and writes the N x N real output.
 *
 *   c2r_probe N in.bin out.bin          in: N*(N/2+1)*2 floats, out: N*N floats
 *
 * Built and run on the GPU box by oracle/fft_probe/make_fixture.py (hipFFTW executes on a GPU); its output is the
 * committed fixture tests/golden/c2r_nonhermitian.npz.  Nothing here is reference code. */
#include <stdio.h>
#include <stdlib.h>
#include <hipfftw.h>   /* the image's FFTW3-API header (AMD hipFFTW) */

int main(int argc, char **argv)
{
    if (argc != 4) return 2;
    int N = atoi(argv[1]);
    size_t nh = (size_t)N * (N / 2 + 1);
    fftwf_complex *in = (fftwf_complex *)fftwf_malloc(sizeof(fftwf_complex) * nh);
    float *out = (float *)fftwf_malloc(sizeof(float) * (size_t)N * N);
    FILE *f = fopen(argv[2], "rb");
    if (!f || fread(in, sizeof(fftwf_complex), nh, f) != nh) return 3;
    fclose(f);
    fftwf_plan p = fftwf_plan_dft_c2r_2d(N, N, in, out, FFTW_ESTIMATE);
    if (!p) return 4;
    /* re-read: planning may overwrite the arrays */
    f = fopen(argv[2], "rb");
    if (!f || fread(in, sizeof(fftwf_complex), nh, f) != nh) return 3;
    fclose(f);
    fftwf_execute_dft_c2r(p, in, out);
    f = fopen(argv[3], "wb");
    if (!f || fwrite(out, sizeof(float), (size_t)N * N, f) != (size_t)N * N) return 5;
    fclose(f);
    fftwf_destroy_plan(p);
    return 0;
}
