// measures the absolute error of v_sin_f32 / v_cos_f32 (argument in revolutions) and v_log_f32 on (0,1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(float* s, float* c, float* l, int n){ int i = blockIdx.x*256+threadIdx.x; if(i>=n) return; float u=(float)(i+1)/(float)n; s[i]=__builtin_amdgcn_sinf(u); c[i]=__builtin_amdgcn_cosf(u); l[i]=__builtin_amdgcn_logf(u)*0.693147182f; }
int main(){ int n=1<<24; float *s,*c,*l; hipMalloc(&s,n*4); hipMalloc(&c,n*4); hipMalloc(&l,n*4); k<<<n/256,256>>>(s,c,l,n); std::vector<float> hs(n),hc(n),hl(n); hipMemcpy(hs.data(),s,n*4,hipMemcpyDeviceToHost); hipMemcpy(hc.data(),c,n*4,hipMemcpyDeviceToHost); hipMemcpy(hl.data(),l,n*4,hipMemcpyDeviceToHost);
 double es=0,ec=0,el=0,ela=0; for(int i=0;i<n;i++){ double u=(double)(float)((float)(i+1)/(float)n); double a=2*M_PI*u; es=fmax(es,fabs(hs[i]-sin(a))); ec=fmax(ec,fabs(hc[i]-cos(a))); double L=log(u); if(L!=0) el=fmax(el,fabs((hl[i]-L)/L)); ela=fmax(ela,fabs(hl[i]-L)); }
 printf("max abs err sin %.3g cos %.3g ; log rel %.3g abs %.3g\n",es,ec,el,ela); return 0; }
