// Host trigonometry of the reference's lines, as its compiler emits it.
//
// g++ (-O1 and above) merges sin(a) and cos(a) of one argument into one call of glibc's
// sincos(a); clang keeps the two calls. glibc's sincos() and cos() do not always agree in the last
// place (a = 0.40005245828215963: the cosines differ by one ulp), and a rotation entry one ulp off
// moves a ray length that sits on a rounding tie (S3P:676, a scan point at a whole number of half
// voxels from the sensor) to the other side of it. The reference is a g++ build (oracle/build_ref.sh
// pins that build, and gcc compiles the oracle the same way), so wherever the reference takes
// both the sine and the cosine of one angle this library asks sincos() for them.
#pragma once
#include <math.h>

struct PfSinCos {
    double s, c;
    explicit PfSinCos(double a) { sincos(a, &s, &c); }
};

// S3P:536-544 (rotate_point's matrix), row-major.
static inline void pf_rot_zyx(double m[9], double z, double y, double x) {
    const PfSinCos Z(z), Y(y), X(x);
    m[0] = Y.c * Z.c;
    m[1] = Z.c * X.s * Y.s - X.c * Z.s;
    m[2] = X.s * Z.s + X.c * Z.c * Y.s;
    m[3] = Y.c * Z.s;
    m[4] = X.c * Z.c + X.s * Y.s * Z.s;
    m[5] = X.c * Y.s * Z.s - Z.c * X.s;
    m[6] = -Y.s;
    m[7] = Y.c * X.s;
    m[8] = X.c * Y.c;
}
