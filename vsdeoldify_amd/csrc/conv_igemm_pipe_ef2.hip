// second half of the compile-time-epilogue kernels (see conv_igemm_pipe_ef.hip): the small tile geometries
#define HAVC_EF_PART 1
#include "conv_igemm_pipe_ef.hip"
