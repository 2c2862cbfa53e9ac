// debug.h -- compile-time switches, same macro names as the reference's debug.h files.
#ifndef SMFFT_DEBUG_H__
#define SMFFT_DEBUG_H__
#ifndef DEBUG
#define DEBUG false
#endif
#define TESTING
#define CUFFT true
#define EXTERNAL true
#define MULTIPLE true
#endif
