// Intentionally empty.  reference test/test_ba.cpp:12 includes
// "utility/simd_library.h" but uses nothing from it (the AVX point warper is
// not on the bundle-adjustment path); this header only
// keeps that include line compiling.
#ifndef BA_FACADE_SIMD_LIBRARY_H_
#define BA_FACADE_SIMD_LIBRARY_H_
#endif
