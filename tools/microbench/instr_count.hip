// Dynamic VALU instruction counts of the RHS pieces: run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES`
// and divide.  Each kernel evaluates one piece once per lane on realistic ascent states.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "../../gelato_amd/csrc/gel_rhs_parts.h"
using namespace gel;
#define K(name, ...) __global__ void name(const double* in, double* out, const double* tabs) { \
  __shared__ double lds[176]; for (int i = threadIdx.x; i < 151; i += blockDim.x) lds[i] = tabs[i]; __syncthreads(); \
  Tables tb{lds, lds+88, lds+115, lds+129, lds+145, 9, 7}; const int t = blockIdx.x * blockDim.x + threadIdx.x; const double* a = in + 16*t; double* o = out + 16*t; __VA_ARGS__ }
K(k_base, o[0]=a[0];)
K(k_div, o[0]=a[0]/a[1];)
K(k_sqrt, o[0]=sqrt(a[0]);)
K(k_sincos, double s,c; sincos(a[9],&s,&c); o[0]=s;o[1]=c;)
K(k_atan2, o[0]=atan2(a[0],a[1]);)
K(k_pow, o[0]=pow(a[10],a[11]);)
K(k_exp, o[0]=exp(-a[10]);)
K(k_acos, o[0]=acos(a[10]-0.5);)
K(k_geolatp, double lat,p,ip; geodetic_lat_p(a[0],a[1],a[2],lat,p,ip); o[0]=lat;o[1]=p;o[2]=ip;)
K(k_fsincos, double s,c; fsincos(a[9],&s,&c); o[0]=s;o[1]=c;)
K(k_flog, o[0]=flog_ratio(a[10]);)
K(k_fdiv, o[0]=fdiv(a[0],a[1]);)
K(k_fsqrt, o[0]=fsqrt(a[0]);)
K(k_fsqrt_rsqrt, double s_,r_; fsqrt_rsqrt(a[0],s_,r_); o[0]=s_;o[1]=r_;)
K(k_atmos, Air p = atmosphere(a[12], tb.atm); o[0]=p.rho;o[1]=p.P;o[2]=p.inv_a;)
K(k_wind, double wn, we; wind_ned2(a[12], tb.wind, tb.winds, tb.Kw, wn, we); o[0]=wn;o[1]=we;)
K(k_interp, o[0]=interp_tab(a[13], tb.ca, tb.cas, tb.Kc, 2, 1);)
K(k_gravity, double r[3]={a[0],a[1],a[2]}; double g[3]; gravity_eci(r,-0.484165371736e-3,g); o[0]=g[0];o[1]=g[1];o[2]=g[2];)
K(k_pos_part, double r[3]={a[0],a[1],a[2]}; PosPart p = pos_part(r, tb, -0.484165371736e-3); o[0]=p.rho;o[1]=p.P;o[2]=p.inv_a;o[3]=p.wn;o[4]=p.we;o[5]=p.g[0];o[6]=p.g[1];o[7]=p.g[2];o[8]=p.shp;o[9]=p.chp;o[10]=p.inv_p;)
K(k_earth, EarthAngle e = earth_angle(a[14]); o[0]=e.c;o[1]=e.s;o[2]=e.ch;o[3]=e.sh;)
K(k_wind_eci, double r[3]={a[0],a[1],a[2]}; EarthAngle e{0.999,0.01,0.9999,0.005}; double w[3]; wind_eci(r,e,0.36,0.93,1.0/4.7e6,10.0,-5.0,w); o[0]=w[0];o[1]=w[1];o[2]=w[2];)
K(k_aero, double r[3]={a[0],a[1],a[2]}; double v[3]={a[3],a[4],a[5]}; EarthAngle e{0.999,0.01,0.9999,0.005}; double w[3]={1,2,3}; double F[3]; aero_force(r,v,0.5,1.0/300.0,e,w,2.21,tb,F); o[0]=F[0];o[1]=F[1];o[2]=F[2];)
K(k_thrustdir, double q[4]={a[5],a[6],a[7],a[8]}; double d[3]; thrust_dir(q,d); o[0]=d[0];o[1]=d[1];o[2]=d[2];)
int main() {
  const int n = 64 * 1024;
  std::vector<double> in(16 * n), tabs(176, 0.0);
  for (int i = 0; i < n; i++) {
    double th = 0.74 + 1e-6 * i, R = 6378137.0 + 10.0 + 1.2 * i;  // altitudes 0..79 km
    double* a = &in[16 * i];
    a[0] = R * cos(th) * 0.8; a[1] = R * cos(th) * 0.6; a[2] = R * sin(th);
    a[3] = 100.0 + 0.05 * i; a[4] = 300.0; a[5] = 50.0; a[6] = 0.5; a[7] = -0.5; a[8] = 0.5; a[9] = 0.7 + 1e-5 * i;
    a[10] = 0.8 + 1e-6 * i; a[11] = 5.2558; a[12] = 1.2 * i; a[13] = 0.1 + 4e-5 * i; a[14] = 0.3;
  }
  // tables: atm (66) | wind 9x3 | ca 7x2, same as the example
  const double lmb[11] = {-0.0065, 0.0, 0.001, 0.0028, 0.0, -0.0028, -0.002, 0.0, 0.0025, 0.012, 0.012};
  const double tmb[11] = {288.15, 216.65, 216.65, 228.65, 270.65, 270.65, 214.65, 186.8673, 186.8673, 240.0, 360.0};
  const double pb[11] = {101325.0, 22632.0, 5474.9, 868.02, 110.91, 66.939, 3.9564, 0.37338, 0.15381, 7.1042e-3, 2.5382e-3};
  for (int k = 0; k < 11; k++) { tabs[k] = lmb[k]; tabs[11+k] = tmb[k]; tabs[22+k] = pb[k]; tabs[33+k] = 8314.32/28.9644;
    tabs[44+k] = fabs(lmb[k]) > 1e-6 ? -9.80665/lmb[k]/tabs[33+k] : 0.0; tabs[55+k] = 9.80665/tabs[33+k]; }
  { const double hb[11] = {0.0, 11000.0, 20000.0, 32000.0, 47000.0, 51000.0, 71000.0, 86000.0, 91000.0, 110000.0, 120000.0}; for (int k = 0; k < 11; k++) tabs[66+k] = hb[k]; }
  const double wind[27] = {-1e8,0,0, 0,0,0, 1000,0,0, 3000,0,10, 11000,0,30, 15000,0,30, 16000,0,25, 23000,0,0, 1e10,0,0};
  const double ca[14] = {0,0.3, 0.7,0.3, 1,0.65, 1.5,0.65, 2,0.6, 5,0.3, 100,0.3};
  for (int k = 0; k < 11; k++) tabs[77+k] = 1.0 / tmb[k];
  for (int i = 0; i < 27; i++) tabs[88+i] = wind[i];
  for (int i = 0; i < 14; i++) tabs[115+i] = ca[i];
  for (int k = 0; k < 8; k++) for (int c = 0; c < 2; c++) tabs[129 + 2*k + c] = (wind[3*(k+1)+1+c] - wind[3*k+1+c]) / (wind[3*(k+1)] - wind[3*k]);
  for (int k = 0; k < 6; k++) tabs[145 + k] = (ca[2*(k+1)+1] - ca[2*k+1]) / (ca[2*(k+1)] - ca[2*k]);
  double *d_in, *d_out, *d_t;
  hipMalloc(&d_in, in.size()*8); hipMalloc(&d_out, in.size()*8); hipMalloc(&d_t, tabs.size()*8);
  hipMemcpy(d_in, in.data(), in.size()*8, hipMemcpyHostToDevice); hipMemcpy(d_t, tabs.data(), tabs.size()*8, hipMemcpyHostToDevice);
#define RUN(k) hipLaunchKernelGGL(k, dim3(n/256), dim3(256), 0, 0, d_in, d_out, d_t);
  RUN(k_base) RUN(k_div) RUN(k_sqrt) RUN(k_sincos) RUN(k_atan2) RUN(k_pow) RUN(k_exp) RUN(k_acos) RUN(k_geolatp) RUN(k_fsincos) RUN(k_flog) RUN(k_fdiv) RUN(k_fsqrt) RUN(k_fsqrt_rsqrt) RUN(k_atmos)
  RUN(k_wind) RUN(k_interp) RUN(k_gravity) RUN(k_pos_part) RUN(k_earth) RUN(k_wind_eci) RUN(k_aero) RUN(k_thrustdir)
  hipDeviceSynchronize();
  printf("done\n");
  return 0;
}
