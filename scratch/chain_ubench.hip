// scratch microbenchmark (not product): where do the cycles of the CTC chain go?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)
constexpr float kNegBig=-1e30f, kLog2e=1.4426950408889634f;
__device__ __forceinline__ float wave_shr1(float v,float fill){return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill),__float_as_int(v),0x138,0xf,0xf,false));}
__device__ __forceinline__ float lse2(float a,float b){float m=fmaxf(a,b);return m+__builtin_amdgcn_logf(__builtin_amdgcn_exp2f(a-m)+__builtin_amdgcn_exp2f(b-m));}
__device__ __forceinline__ float lse3(float a,float b,float c){float m=fmaxf(fmaxf(a,b),c);return m+__builtin_amdgcn_logf(__builtin_amdgcn_exp2f(a-m)+__builtin_amdgcn_exp2f(b-m)+__builtin_amdgcn_exp2f(c-m));}
template<int MODE> // 0 full, 1 no store, 2 no load, 3 neither, 4 prob-domain(no renorm) full io
__global__ void __launch_bounds__(64) k(const float* __restrict__ x,int T,int C,int L,float2* __restrict__ out,long long* cyc){
  const int b=blockIdx.x,lane=threadIdx.x; const int col=lane<L? (lane*7+b)%(C-1):C-1; const bool skip=lane>=1&&lane<L&&(lane%3);
  const float* xb=x+(long)b*T*C; float2* o=out+(long)b*T*64;
  float ab=lane==0?0.f:kNegBig, al=kNegBig; if(MODE==4){ab=lane==0?1.f:0.f; al=0.f;}
  constexpr int D=16; float ring[D];
  #pragma unroll
  for(int j=0;j<D;++j) ring[j]= (MODE==2||MODE==3)? 0.01f*j : xb[(long)j*C+col];
  long long t0=clock64();
  for(int c=0;c<T/D;++c){
    #pragma unroll
    for(int j=0;j<D;++j){ int step=c*D+j; float raw=ring[j]; int sn=min(step+D,T-1);
      if(MODE==2||MODE==3) ring[j]=raw+1e-3f; else ring[j]=xb[(long)sn*C+col];
      if(MODE==4){
        float e=__builtin_amdgcn_exp2f(raw*kLog2e-2.0f); float eb=__int_as_float(__builtin_amdgcn_readlane(__float_as_int(e),L));
        float pal=wave_shr1(al,0.f); float nb=(ab+pal)*eb; float nl=(al+ab+(skip?pal:0.f))*e; ab=nb; al=nl;
        o[(long)step*64+lane]=make_float2(ab,al);
      } else {
      float xs=raw*kLog2e; xs=xs>kNegBig?xs:kNegBig; float xbl=__int_as_float(__builtin_amdgcn_readlane(__float_as_int(xs),L));
      float pal=wave_shr1(al,kNegBig); float nb=lse2(ab,pal); float nl=lse3(al,ab,skip?pal:kNegBig);
      ab=fmaxf(nb+xbl,kNegBig); al=fmaxf(nl+(lane<L?xs:kNegBig),kNegBig);
      if(MODE==0||MODE==2){ if(lane<=L) o[(long)step*64+lane]=make_float2(ab,al);} }
    }
  }
  long long t1=clock64();
  if(lane==0) cyc[b]=t1-t0;
  if(MODE==1||MODE==3){ if(ab+al==123.f) o[0]=make_float2(ab,al);}
}
template<int MODE> void run(const char* name,const float* x,int B,int T,int C,int L,float2* out,long long* cyc){
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for(int w=0;w<3;++w) hipLaunchKernelGGL(k<MODE>,dim3(B),dim3(64),0,0,x,T,C,L,out,cyc);
  CK(hipEventRecord(e0)); const int R=20;
  for(int r=0;r<R;++r) hipLaunchKernelGGL(k<MODE>,dim3(B),dim3(64),0,0,x,T,C,L,out,cyc);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  std::vector<long long> h(B); CK(hipMemcpy(h.data(),cyc,B*8,hipMemcpyDeviceToHost)); double avg=0; long long mx=0; for(auto v:h){avg+=v; if(v>mx)mx=v;} avg/=B;
  printf("%-28s B=%d T=%d: %.1f us/launch, %.1f ns/step, clock64 avg %.0f max %lld ticks/step=%.1f\n",name,B,T,ms*1e3/R,ms*1e6/R/T,avg,mx,avg/T);
}
int main(){ int C=100,L=44; int Tmax=4000,Bmax=512; float* x; float2* out; long long* cyc;
  CK(hipMalloc(&x,(size_t)Bmax*Tmax*C*4)); CK(hipMalloc(&out,(size_t)Bmax*Tmax*64*8)); CK(hipMalloc(&cyc,Bmax*8));
  std::vector<float> h((size_t)Bmax*Tmax*C); for(auto&v:h) v=(rand()/(float)RAND_MAX-0.5f)*4; CK(hipMemcpy(x,h.data(),h.size()*4,hipMemcpyHostToDevice));
  for(int T: {1000,2000}) for(int B: {64,256,512}){ run<0>("log full",x,B,T,C,L,out,cyc); }
  run<1>("log nostore",x,256,1000,C,L,out,cyc); run<2>("log noload",x,256,1000,C,L,out,cyc); run<3>("log compute only",x,256,1000,C,L,out,cyc);
  run<3>("log compute only B=64",x,64,1000,C,L,out,cyc);
  run<4>("prob full io",x,256,1000,C,L,out,cyc); run<4>("prob full io B=64",x,64,1000,C,L,out,cyc);
  return 0; }
