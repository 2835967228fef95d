// developer bisect tool (not shipped): runs pieces of the device pose math one at a time
#include "../../visual_odom_amd/csrc/vo_epnp.h"
#include <stdio.h>
#include <stdlib.h>
using namespace vo;

__global__ void k_svd3(const double *in, double *out)
{
    double At[9], w[3], vt[9];
    for (int i = 0; i < 9; i++) At[i] = in[i];
    jacobi_svd<3, 3, true>(At, w, vt);
    for (int i = 0; i < 3; i++) out[i] = w[i];
}
__global__ void k_inv3(const double *in, double *out)
{
    double A[9], Ai[9];
    for (int i = 0; i < 9; i++) A[i] = in[i];
    invert_svd<3>(A, Ai);
    for (int i = 0; i < 9; i++) out[i] = Ai[i];
}
__global__ void k_svd12(const double *in, double *out)
{
    double At[144], w[12];
    for (int i = 0; i < 144; i++) At[i] = in[i];
    jacobi_svd<12, 12, false>(At, w, nullptr);
    for (int i = 0; i < 12; i++) out[i] = w[i];
}
__global__ void k_solve64(const double *in, double *out)
{
    double A[24], b[6], x[4];
    for (int i = 0; i < 24; i++) A[i] = in[i];
    for (int i = 0; i < 6; i++) b[i] = in[24 + i];
    solve_svd<6, 4>(A, b, x);
    for (int i = 0; i < 4; i++) out[i] = x[i];
}
__global__ void k_qr(const double *in, double *out)
{
    double A[24], b[6], x[4] = {0, 0, 0, 0};
    for (int i = 0; i < 24; i++) A[i] = in[i];
    for (int i = 0; i < 6; i++) b[i] = in[24 + i];
    qr_solve_6x4(A, b, x);
    for (int i = 0; i < 4; i++) out[i] = x[i];
}
__global__ void k_m2v(const double *in, double *out)
{
    double R[9], r[3];
    for (int i = 0; i < 9; i++) R[i] = in[i];
    rodrigues_m2v(R, r);
    for (int i = 0; i < 3; i++) out[i] = r[i];
}
__global__ void k_gn(const double *in, double *out)
{
    double L[60], rho[6], betas[4];
    for (int i = 0; i < 60; i++) L[i] = in[i];
    for (int i = 0; i < 6; i++) rho[i] = in[60 + i] + 2.0;
    for (int i = 0; i < 4; i++) betas[i] = in[70 + i];
    epnp_gauss_newton(L, rho, betas);
    for (int i = 0; i < 4; i++) out[i] = betas[i];
}
__global__ void k_solve65(const double *in, double *out)
{
    double A[30], b[6], x[5];
    for (int i = 0; i < 30; i++) A[i] = in[i];
    for (int i = 0; i < 6; i++) b[i] = in[30 + i];
    solve_svd<6, 5>(A, b, x);
    for (int i = 0; i < 5; i++) out[i] = x[i];
    double A3[18], x3[3];
    for (int i = 0; i < 18; i++) A3[i] = in[40 + i];
    solve_svd<6, 3>(A3, b, x3);
    for (int i = 0; i < 3; i++) out[5 + i] = x3[i];
}
__global__ void k_epnp(const float *xyz, const float *uv, const float *K, double *out)
{
    float x5[15], u5[10], Kf[9];
    for (int i = 0; i < 15; i++) x5[i] = xyz[i];
    for (int i = 0; i < 10; i++) u5[i] = uv[i];
    for (int i = 0; i < 9; i++) Kf[i] = K[i];
    double rv[3], tv[3];
    epnp5_solve(x5, u5, Kf, rv, tv);
    for (int i = 0; i < 3; i++) { out[i] = rv[i]; out[3 + i] = tv[i]; }
}

int main(int argc, char **argv)
{
    int which = argc > 1 ? atoi(argv[1]) : 0;
    double h_in[256], h_out[16] = {0};
    srand(1);
    for (int i = 0; i < 256; i++) h_in[i] = (rand() % 2000) / 1000.0 - 1.0;
    if (which == 2) { // symmetric PSD rank 10 like MtM
        double M[120];
        for (int i = 0; i < 120; i++) M[i] = h_in[i];
        for (int i = 0; i < 12; i++) for (int j = 0; j < 12; j++) { double s = 0; for (int k = 0; k < 10; k++) s += M[k*12+i]*M[k*12+j]; h_in[i*12+j] = s; }
    }
    if (which == 5) { double R[9] = {0.9998,-0.01,0.02, 0.01,0.9999,0.005, -0.02,-0.005,0.9998}; for (int i=0;i<9;i++) h_in[i]=R[i]; }
    double *d_in, *d_out; float *d_f;
    hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, sizeof(h_out)); hipMalloc(&d_f, 64 * sizeof(float));
    hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    float hf[64] = { 1,0.5f,10,  -3,1,15,  4,-1,8,  -2,0.2f,20,  0.5f,1.2f,12 };
    float K[9] = {718.856f,0,607.1928f, 0,718.856f,185.2157f, 0,0,1};
    for (int i = 0; i < 5; i++) { hf[16+2*i] = K[0]*(hf[3*i]+0.1f)/(hf[3*i+2]-0.9f)+K[2]; hf[16+2*i+1] = K[4]*(hf[3*i+1])/(hf[3*i+2]-0.9f)+K[5]; }
    for (int i = 0; i < 9; i++) hf[32+i] = K[i];
    hipMemcpy(d_f, hf, sizeof(hf), hipMemcpyHostToDevice);
    switch (which) {
    case 0: k_svd3<<<1, 1>>>(d_in, d_out); break;
    case 1: k_inv3<<<1, 1>>>(d_in, d_out); break;
    case 2: k_svd12<<<1, 1>>>(d_in, d_out); break;
    case 3: k_solve64<<<1, 1>>>(d_in, d_out); break;
    case 4: k_qr<<<1, 1>>>(d_in, d_out); break;
    case 5: k_m2v<<<1, 1>>>(d_in, d_out); break;
    case 7: k_gn<<<1, 1>>>(d_in, d_out); break;
    case 8: k_solve65<<<1, 1>>>(d_in, d_out); break;
    case 6: k_epnp<<<1, 64>>>(d_f, d_f + 16, d_f + 32, d_out); break;
    }
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("case %d: %s :", which, hipGetErrorString(e));
    for (int i = 0; i < 12; i++) printf(" %.6g", h_out[i]);
    printf("\n");
    // host reference of the same code
    if (which == 6) { double rv[3], tv[3]; epnp5_solve(hf, hf+16, K, rv, tv); printf("host  6: %.9g %.9g %.9g %.9g %.9g %.9g\n", rv[0],rv[1],rv[2],tv[0],tv[1],tv[2]); }
    if (which == 2) { double At[144], w[12]; for (int i=0;i<144;i++) At[i]=h_in[i]; jacobi_svd<12,12,false>(At,w,nullptr); printf("host  2:"); for (int i=0;i<12;i++) printf(" %.6g", w[i]); printf("\n"); }
    return 0;
}
