// third part of the compile-time-epilogue kernels (see conv_igemm_pipe_ef.hip): the 96- and 192-row tile geometries
#define HAVC_EF_PART 2
#include "conv_igemm_pipe_ef.hip"
