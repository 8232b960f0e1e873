// tools/membench.hip — streaming-bandwidth probes on MI355X used to size the stencil kernels.
// Build: hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s line %d\n",hipGetErrorString(e),__LINE__);exit(1);} }while(0)

template<int U, bool NT>
__global__ void __launch_bounds__(256) copy_k(const f4* __restrict__ a, f4* __restrict__ c, long n){
  const long stride=(long)gridDim.x*blockDim.x;
  long q=(long)blockIdx.x*blockDim.x+threadIdx.x;
  for(; q + (U-1)*stride < n; q += U*stride){
    f4 v[U];
#pragma unroll
    for(int u=0;u<U;++u) v[u]= NT ? __builtin_nontemporal_load(a+q+u*stride) : a[q+u*stride];
#pragma unroll
    for(int u=0;u<U;++u){ if(NT) __builtin_nontemporal_store(v[u], c+q+u*stride); else c[q+u*stride]=v[u]; }
  }
  for(; q<n; q+=stride) c[q]=a[q];
}
// 2 reads + 1 write (the Jacobi mix), no reuse
template<int U, bool NT>
__global__ void __launch_bounds__(256) triad_k(const f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ c, long n){
  const long stride=(long)gridDim.x*blockDim.x;
  long q=(long)blockIdx.x*blockDim.x+threadIdx.x;
  for(; q + (U-1)*stride < n; q += U*stride){
    f4 v[U], w[U];
#pragma unroll
    for(int u=0;u<U;++u){ v[u]=a[q+u*stride]; w[u]=b[q+u*stride]; }
#pragma unroll
    for(int u=0;u<U;++u){ f4 r = v[u]+w[u];
      if(NT) __builtin_nontemporal_store(r, c+q+u*stride); else c[q+u*stride]=r; }
  }
  for(; q<n; q+=stride){ f4 r=a[q]+b[q]; if(NT) __builtin_nontemporal_store(r,c+q); else c[q]=r; }
}
// block-contiguous variant: each block streams a contiguous chunk
template<int U>
__global__ void __launch_bounds__(256) copy_chunk_k(const f4* __restrict__ a, f4* __restrict__ c, long n){
  const long per=(n+gridDim.x-1)/gridDim.x;
  const long b0=(long)blockIdx.x*per, b1 = b0+per<n? b0+per:n;
  for(long q=b0+threadIdx.x; q<b1; q+=256*U){
#pragma unroll
    for(int u=0;u<U;++u){ long p=q+u*256; if(p<b1) c[p]=a[p]; }
  }
}
template<class F> float timeit(F f,int reps){ hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); f(); CK(hipDeviceSynchronize()); float best=1e30f; for(int r=0;r<reps;++r){ CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1)); if(ms<best)best=ms;} return best; }
int main(int argc,char**argv){
  size_t bytes = (argc>1? atol(argv[1]) : 1024)*(size_t)(1<<20);
  long n=bytes/16; f4 *a,*b,*c; CK(hipMalloc(&a,bytes)); CK(hipMalloc(&b,bytes)); CK(hipMalloc(&c,bytes));
  CK(hipMemset(a,1,bytes)); CK(hipMemset(b,2,bytes)); CK(hipMemset(c,0,bytes)); CK(hipDeviceSynchronize());
  int grids[]={1024,2048,4096,8192,16384,65536, (int)((n+255)/256)};
  for(int g: grids){
    float t1=timeit([&]{ hipLaunchKernelGGL((copy_k<1,false>),dim3(g),dim3(256),0,0,a,c,n);},5);
    float t4=timeit([&]{ hipLaunchKernelGGL((copy_k<4,false>),dim3(g),dim3(256),0,0,a,c,n);},5);
    float t4n=timeit([&]{ hipLaunchKernelGGL((copy_k<4,true>),dim3(g),dim3(256),0,0,a,c,n);},5);
    float tc=timeit([&]{ hipLaunchKernelGGL((copy_chunk_k<4>),dim3(g),dim3(256),0,0,a,c,n);},5);
    float tt=timeit([&]{ hipLaunchKernelGGL((triad_k<2,false>),dim3(g),dim3(256),0,0,a,b,c,n);},5);
    float ttn=timeit([&]{ hipLaunchKernelGGL((triad_k<2,true>),dim3(g),dim3(256),0,0,a,b,c,n);},5);
    printf("grid %8d: copy U1 %.0f  U4 %.0f  U4nt %.0f  chunk %.0f GB/s | triad %.0f  triad-nt %.0f GB/s\n", g,
      2.0*bytes/t1/1e6, 2.0*bytes/t4/1e6, 2.0*bytes/t4n/1e6, 2.0*bytes/tc/1e6, 3.0*bytes/tt/1e6, 3.0*bytes/ttn/1e6);
  }
  float tm=timeit([&]{ CK(hipMemcpyAsync(c,a,bytes,hipMemcpyDeviceToDevice,0)); },5);
  printf("hipMemcpy D2D %.0f GB/s\n", 2.0*bytes/tm/1e6);
  return 0;
}
