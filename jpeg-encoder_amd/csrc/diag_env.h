// diag_env.h — diagnostic switches.  Alternative code paths the tests force and the A/B knobs of tools/diag are read from
// the environment ONLY in -DJPEGENC_DIAG builds (libjpegenc_mi355x_diag.so: built by build.sh beside the shipping library
// and loaded - JPEGENC_LIB - by the tests and tools that need a switch).  The shipping library reads two variables,
// JPEGENC_TRACE (per-frame stage times on stderr) and JPEGENC_NUMA_BIND (default of jpegenc_encoder_set_numa_bind), and
// behaves the same in every environment otherwise.
#pragma once
#include <stdlib.h>

#ifdef JPEGENC_DIAG
#define JPEGENC_DIAG_ENV(name) getenv(name)
#else
#define JPEGENC_DIAG_ENV(name) ((const char *)nullptr)
#endif
