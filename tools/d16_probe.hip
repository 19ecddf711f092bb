// d16_probe.hip — does a D16 LDS load keep or clear the other half of its destination register on this part?  (gfx950, ECC on)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(uint32_t* out)
{
	__shared__ uint32_t lds[64];
	lds[threadIdx.x] = 0x44332211u + threadIdx.x;
	__syncthreads();
	uint32_t r = 0xAABBCCDDu, r2 = 0xAABBCCDDu, a = threadIdx.x * 4;
	asm volatile("ds_read_u8_d16_hi %0, %2\n ds_read_u8_d16 %1, %2 offset:1\n s_waitcnt lgkmcnt(0)" : "+v"(r), "+v"(r2) : "v"(a));
	uint32_t r3;
	asm volatile("ds_read_u8 %0, %1 offset:2\n s_waitcnt lgkmcnt(0)\n ds_read_u8_d16_hi %0, %1 offset:3\n s_waitcnt lgkmcnt(0)" : "=&v"(r3) : "v"(a));
	out[threadIdx.x * 3] = r; out[threadIdx.x * 3 + 1] = r2; out[threadIdx.x * 3 + 2] = r3;
}
int main()
{
	uint32_t* d; hipMalloc(&d, 64 * 12); hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
	uint32_t h[192]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
	printf("lane0: word 0x44332211; init 0xAABBCCDD\n d16_hi(byte0) -> %08x (keeps: aa11ccdd? clears: 00110000)\n d16(byte1) -> %08x (keeps: aabb0022, clears 00000022)\n u8(byte2) then d16_hi(byte3) -> %08x (keeps: 00440033, clears: 00440000)\n", h[0], h[1], h[2]);
	return 0;
}
