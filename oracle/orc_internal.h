/* orc_internal.h -- CPU ORACLE internals (test infrastructure, NOT product code). */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H

#include "vo_oracle.h"

#include <stdlib.h>

#define ORC_SVD_MAXN 12
#define ORC_SVD_MAXM 12

uint32_t orc_rng_next(uint64_t *state);
void orc_jacobi_svd(double *At, int astep, double *W, double *Vt, int vstep, int m, int n, int n1);
void orc_solve_svd(const double *A, int m, int n, const double *b, double *x);
void orc_invert_svd(const double *A, int n, double *Ainv);
void orc_project_points_d(const double *M, int n, const double *rvec, const double *tvec,
                          const double *A4, double *m, double *dpdr, double *dpdt, int jstride);
/* EPnP on double object points / pixel image points; fu,fv,uc,vc doubles */
void orc_epnp_d(const double *pws, const double *us, int n, double fu, double fv, double uc,
                double vc, double R[3][3], double t[3]);

#endif
