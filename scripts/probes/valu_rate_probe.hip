#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE> __device__ __forceinline__ uint32_t op3(uint32_t x, uint32_t g)
{
  if (MODE == 0) { s16x2 a=__builtin_bit_cast(s16x2,x), b=__builtin_bit_cast(s16x2,g); a=__builtin_elementwise_add_sat(a,b); u16x2 c=__builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2,a),__builtin_bit_cast(u16x2,g)); a=__builtin_elementwise_max(__builtin_bit_cast(s16x2,c), __builtin_bit_cast(s16x2,x)); return __builtin_bit_cast(uint32_t,a); }
  if (MODE == 1) { int a=(int)x+(int)g; a = a - (int)(g>>1); a = a > (int)x ? a : (int)x; return (uint32_t)a; }
  if (MODE == 2) { h2 a=__builtin_bit_cast(h2,x), b=__builtin_bit_cast(h2,g); a = a + b; a = a - b; a = __builtin_elementwise_max(a, __builtin_bit_cast(h2,x)); return __builtin_bit_cast(uint32_t,a); }
  if (MODE == 3) { uint32_t r; asm volatile("v_max3_i32 %0, %1, %2, %1\n v_add3_u32 %0, %0, %2, %1\n v_max3_i32 %0, %0, %2, %1" : "=v"(r) : "v"(x), "v"(g)); return r; }
  if (MODE == 4) { uint32_t r; asm volatile("v_pk_maximum3_f16 %0, %1, %2, %1\n v_pk_add_f16 %0, %0, %2\n v_pk_maximum3_f16 %0, %0, %2, %1" : "=v"(r) : "v"(x), "v"(g)); return r; }
  if (MODE == 5) { float a=__builtin_bit_cast(float,x), b=__builtin_bit_cast(float,g); a = a*b+b; a=a*b+b; a=a*b+b; return __builtin_bit_cast(uint32_t,a); }
  if (MODE == 6) { uint32_t r; asm volatile("v_pk_max_i16 %0, %1, %2\n v_pk_max_i16 %0, %0, %2\n v_pk_max_i16 %0, %0, %1" : "=v"(r) : "v"(x), "v"(g)); return r; }
  if (MODE == 7) { uint32_t r; asm volatile("v_pk_add_u16 %0, %1, %2\n v_pk_add_u16 %0, %0, %2\n v_pk_add_u16 %0, %0, %1" : "=v"(r) : "v"(x), "v"(g)); return r; }
  if (MODE == 8) { uint32_t r; asm volatile("v_pk_max_f16 %0, %1, %2\n v_pk_max_f16 %0, %0, %2\n v_pk_max_f16 %0, %0, %1" : "=v"(r) : "v"(x), "v"(g)); return r; }
  if (MODE == 9) { uint32_t r; asm volatile("v_max_i32 %0, %1, %2\n v_max_i32 %0, %0, %2\n v_max_i32 %0, %0, %1" : "=v"(r) : "v"(x), "v"(g)); return r; }
  if (MODE == 10) { uint32_t r; asm volatile("v_max_u16 %0, %1, %2\n v_max_u16 %0, %0, %2\n v_add_u16 %0, %0, %1" : "=v"(r) : "v"(x), "v"(g)); return r; }
  return x;
}
template <int MODE> __global__ void __launch_bounds__(256) k(uint32_t* sink, int iters, uint32_t g)
{
  uint32_t gid = blockIdx.x*blockDim.x+threadIdx.x;
  uint32_t x0=gid,x1=gid*3,x2=gid*5,x3=gid*7,x4=gid*11,x5=gid*13,x6=gid*17,x7=gid*19;
  for (int it=0; it<iters; ++it) {
#pragma unroll
    for (int k2=0;k2<4;++k2){ x0=op3<MODE>(x0,g);x1=op3<MODE>(x1,g);x2=op3<MODE>(x2,g);x3=op3<MODE>(x3,g);x4=op3<MODE>(x4,g);x5=op3<MODE>(x5,g);x6=op3<MODE>(x6,g);x7=op3<MODE>(x7,g);}
  }
  sink[gid]=x0^x1^x2^x3^x4^x5^x6^x7;
}
template <int MODE> void run(const char* name, uint32_t* sink)
{
  hipEvent_t a,b; hipEventCreate(&a); hipEventCreate(&b);
  const int blocks=8192, iters=2000;
  k<MODE><<<blocks,256>>>(sink, 10, 0x00030001u); hipDeviceSynchronize();
  float best=1e9;
  for (int r=0;r<3;++r){ hipEventRecord(a); k<MODE><<<blocks,256>>>(sink, iters, 0x00030001u); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms,a,b); if(ms<best)best=ms; }
  double ops=(double)blocks*256*iters*96.0;
  printf("%-28s %8.2f ms  %7.2f T lane-instr/s\n", name, best, ops/(best*1e-3)/1e12);
}
int main(){ uint32_t* sink; hipMalloc(&sink, 8192*256*4);
 run<0>("pk add_i16/sub_u16/max_i16", sink); run<6>("v_pk_max_i16 x3", sink); run<7>("v_pk_add_u16 x3", sink); run<1>("i32 add/sub/max", sink); run<9>("v_max_i32 x3", sink);
 run<2>("pk f16 add/sub/max", sink); run<8>("v_pk_max_f16 x3", sink); run<4>("pk_maximum3_f16/add", sink); run<3>("max3_i32/add3", sink); run<5>("fma_f32 x3", sink); run<10>("v_max_u16/v_add_u16", sink);
 return 0; }
