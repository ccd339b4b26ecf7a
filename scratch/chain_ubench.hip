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

// ---- variant E: extended-exponent floats, separate exponent per state, no transcendental on the chain
template<int IO>
__global__ void __launch_bounds__(64) kE(const float* __restrict__ x,int T,int C,int L,float2* __restrict__ out,long long* cyc){
  const int b=blockIdx.x,lane=threadIdx.x; const int col=lane<L? (lane*7+b)%(C-1):C-1; const bool skip=lane>=1&&lane<L&&(lane%3);
  const float* xb=x+(long)b*T*C; float2* o=out+(long)b*T*64;
  const int EMPTY=-(1<<28);
  float pb=lane==0?1.f:0.f, pl=0.f; int eb=lane==0?0:EMPTY, el=EMPTY;
  constexpr int D=16; float ring[D];
  #pragma unroll
  for(int j=0;j<D;++j) ring[j]= IO? xb[(long)j*C+col] : 0.01f*j;
  long long t0=clock64();
  for(int c=0;c<T/D;++c){
    #pragma unroll
    for(int j=0;j<D;++j){ int step=c*D+j; float raw=ring[j]; int sn=min(step+D,T-1);
      if(IO) ring[j]=xb[(long)sn*C+col]; else ring[j]=raw+1e-3f;
      float xs=raw*kLog2e; float nf=floorf(xs); float m=__builtin_amdgcn_exp2f(xs-nf); int n=(int)nf;
      float mb=__int_as_float(__builtin_amdgcn_readlane(__float_as_int(m),L)); int nb=__builtin_amdgcn_readlane(n,L);
      float ql=wave_shr1(pl,0.f); int qe=__builtin_amdgcn_update_dpp(EMPTY,el,0x138,0xf,0xf,false);
      int em=max(eb,qe); float sb=ldexpf(pb,eb-em)+ldexpf(ql,qe-em);
      int qe2=skip?qe:EMPTY; int em3=max(max(el,eb),qe2); float sl=ldexpf(pl,el-em3)+ldexpf(pb,eb-em3)+ldexpf(ql,qe2-em3);
      pb=sb*mb; eb=em+nb; pl=sl*m; el=em3+n;
      if((j&7)==7){ int k=__builtin_amdgcn_frexp_expf(pb); pb=__builtin_amdgcn_frexp_mantf(pb); eb+=k; k=__builtin_amdgcn_frexp_expf(pl); pl=__builtin_amdgcn_frexp_mantf(pl); el+=k; }
      if(IO){ if(lane<=L) o[(long)step*64+lane]=make_float2(pb,pl);} }
  }
  long long t1=clock64(); if(lane==0) cyc[b]=t1-t0;
  if(pb+pl+eb+el==123.f) o[0]=make_float2(pb,pl);
}
// ---- variant P4: per-lane block exponent (pair-shared), renorm every 4 frames
template<int IO>
__global__ void __launch_bounds__(64) kP(const float* __restrict__ x,int T,int C,int L,float2* __restrict__ out,long long* cyc){
  const int b=blockIdx.x,lane=threadIdx.x; const int col=lane<L? (lane*7+b)%(C-1):C-1; const bool skip=lane>=1&&lane<L&&(lane%3);
  const float* xb=x+(long)b*T*C; float2* o=out+(long)b*T*64;
  float pb=lane==0?1.f:0.f, pl=0.f; int e=0; float f=lane==0?0.f:1.f; int bad=0;
  constexpr int D=16; float ring[D];
  #pragma unroll
  for(int j=0;j<D;++j) ring[j]= IO? xb[(long)j*C+col] : 0.01f*j;
  long long t0=clock64();
  for(int c=0;c<T/D;++c){
    #pragma unroll
    for(int g=0;g<D/4;++g){
      float xs[4], xbs[4];
      #pragma unroll
      for(int j=0;j<4;++j){ int step=c*D+g*4+j; float raw=ring[g*4+j]; int sn=min(step+D,T-1);
        if(IO) ring[g*4+j]=xb[(long)sn*C+col]; else ring[g*4+j]=raw+1e-3f;
        xs[j]=raw*kLog2e; xbs[j]=__int_as_float(__builtin_amdgcn_readlane(__float_as_int(xs[j]),L)); }
      float r=fmaxf(fmaxf(fmaxf(xs[0],xs[1]),fmaxf(xs[2],xs[3])),fmaxf(fmaxf(xbs[0],xbs[1]),fmaxf(xbs[2],xbs[3]))); r=rintf(r);
      #pragma unroll
      for(int j=0;j<4;++j){ int step=c*D+g*4+j;
        float fl=__builtin_amdgcn_exp2f(xs[j]-r), fb=__builtin_amdgcn_exp2f(xbs[j]-r);
        float q=wave_shr1(pl,0.f)*f; float nb=(pb+q)*fb; float nl=(pl+pb+(skip?q:0.f))*fl; pb=nb; pl=nl;
        if(IO){ if(lane<=L) o[(long)step*64+lane]=make_float2(pb,pl);} }
      // renorm
      float mx=fmaxf(pb,pl); bad|=(mx>0.f&&mx<1e-24f)||(mx>1e30f);
      int k=__builtin_amdgcn_frexp_expf(mx); pb=ldexpf(pb,-k); pl=ldexpf(pl,-k); e+=k+4*(int)r;
      int qe=__builtin_amdgcn_update_dpp(e,e,0x138,0xf,0xf,false); int d=qe-e; int sh=max(d-40,0);
      pb=ldexpf(pb,-sh); pl=ldexpf(pl,-sh); e+=sh; d-=sh; f=lane==0?0.f:ldexpf(1.f,d);
    }
  }
  long long t1=clock64(); if(lane==0) cyc[b]=t1-t0;
  if(pb+pl+e+bad==123.f) o[0]=make_float2(pb,pl);
}
template<class K> void runk(K kern,const char* name,const float* x,int B,int T,int C,int L,float2* out,long long* cyc){
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for(int w=0;w<3;++w) hipLaunchKernelGGL(kern,dim3(B),dim3(64),0,0,x,T,C,L,out,cyc);
  CK(hipEventRecord(e0)); const int R=20;
  for(int r=0;r<R;++r) hipLaunchKernelGGL(kern,dim3(B),dim3(64),0,0,x,T,C,L,out,cyc);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1));
  std::vector<long long> h(B); CK(hipMemcpy(h.data(),cyc,B*8,hipMemcpyDeviceToHost)); double avg=0; for(auto v:h) avg+=v; avg/=B;
  printf("%-28s B=%d T=%d: %.1f us/launch, ticks/step=%.1f\n",name,B,T,ms*1e3/R,avg/T);
}

int main(){ int C=100,L=44; int Tmax=4000,Bmax=512; float* x; float2* out; long long* cyc;
  CK(hipMalloc(&x,(size_t)Bmax*Tmax*C*4)); CK(hipMalloc(&out,(size_t)Bmax*Tmax*64*8)); CK(hipMalloc(&cyc,Bmax*8));
  std::vector<float> h((size_t)Bmax*Tmax*C); for(auto&v:h) v=(rand()/(float)RAND_MAX-0.5f)*4; CK(hipMemcpy(x,h.data(),h.size()*4,hipMemcpyHostToDevice));
  for(int T: {1000,2000}) for(int B: {64,256,512}){ run<0>("log full",x,B,T,C,L,out,cyc); }
  run<1>("log nostore",x,256,1000,C,L,out,cyc); run<2>("log noload",x,256,1000,C,L,out,cyc); run<3>("log compute only",x,256,1000,C,L,out,cyc);
  run<3>("log compute only B=64",x,64,1000,C,L,out,cyc);
  run<4>("prob full io",x,256,1000,C,L,out,cyc); run<4>("prob full io B=64",x,64,1000,C,L,out,cyc);
  runk(kE<0>,"EEF compute only",x,256,1000,C,L,out,cyc); runk(kE<1>,"EEF full io",x,256,1000,C,L,out,cyc);
  runk(kP<0>,"P4 compute only",x,256,1000,C,L,out,cyc); runk(kP<1>,"P4 full io",x,256,1000,C,L,out,cyc);
  return 0; }
