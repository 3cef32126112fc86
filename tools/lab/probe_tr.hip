// Hardware probe (dev tool, not product): dumps the lane mapping of ds_read_b64_tr_b16 and checks
// the operand/accumulator layout of v_mfma_f32_32x32x16_bf16 against a host reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void k_tr(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[1024];
  int l = threadIdx.x;
  for (int i = l; i < 1024; i += 64) lds[i] = (short)i;
  __syncthreads();
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + l * 4));
  for (int j = 0; j < 4; j++) out[l * 4 + j] = t[j];
}
// canonical use: a [16 k][32 n] row-major tile (row stride 32 elems); each 16-lane group g reads
// rows 4*(g&1).. and columns 16*(g>>1)..: lane i -> &tile[4*?][...]
__global__ void k_tr2(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[16 * 64];
  int l = threadIdx.x;
  for (int i = l; i < 16 * 64; i += 64) lds[i] = (short)i;   // value = k*64 + n
  __syncthreads();
  int i = l & 15, g = l >> 4;
  // group g: columns 16*(g&1).. of rows 4*(g>>1)..   lane i: row (i>>2), 4 cols at 4*(i&3)
  int row = 4 * (g >> 1) + (i >> 2), col = 16 * (g & 1) + 4 * (i & 3);
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + row * 64 + col));
  for (int j = 0; j < 4; j++) out[l * 4 + j] = t[j];
}
__device__ inline unsigned short f2bf(float f) { unsigned u = __float_as_uint(f); u += 0x7fff + ((u >> 16) & 1); return u >> 16; }
__global__ void k_mfma(const unsigned short* A, const unsigned short* B, float* D) {
  // A: [32][16] row-major (i,k), B: [16][32] row-major (k,j)
  int l = threadIdx.x;
  s16x8 a, b;
  for (int e = 0; e < 8; e++) { a[e] = A[(l & 31) * 16 + 8 * (l >> 5) + e]; b[e] = B[(8 * (l >> 5) + e) * 32 + (l & 31)]; }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  for (int r = 0; r < 16; r++) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
static unsigned short h_f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return u >> 16; }
static float h_bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  std::vector<short> h(256);
  k_tr<<<1, 64>>>(d); hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
  printf("TR1 (addr = lane*4 elems): out[lane][j] = source element index\n");
  for (int l = 0; l < 64; l++) { printf("l%02d:", l); for (int j = 0; j < 4; j++) printf(" %4d", h[l * 4 + j]); printf(l % 4 == 3 ? "\n" : "   "); }
  k_tr2<<<1, 64>>>(d); hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
  printf("TR2 (tile [16k][64n], value=k*64+n): out[lane][j] -> (k,n)\n");
  for (int l = 0; l < 64; l++) { printf("l%02d:", l); for (int j = 0; j < 4; j++) printf(" (%2d,%2d)", h[l * 4 + j] / 64, h[l * 4 + j] % 64); printf(l % 2 == 1 ? "\n" : "   "); }
  // mfma check
  std::vector<unsigned short> A(32 * 16), B(16 * 32); std::vector<float> D(32 * 32), R(32 * 32, 0.f);
  srand(1);
  for (auto& x : A) x = h_f2bf((rand() % 17 - 8) / 4.0f);
  for (auto& x : B) x = h_f2bf((rand() % 13 - 6) / 2.0f);
  for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) for (int k = 0; k < 16; k++) R[i * 32 + j] += h_bf2f(A[i * 16 + k]) * h_bf2f(B[k * 32 + j]);
  unsigned short *dA, *dB; float* dD; hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, D.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  k_mfma<<<1, 64>>>(dA, dB, dD); hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  double err = 0; for (int i = 0; i < 1024; i++) err = fmax(err, fabs(D[i] - R[i]));
  printf("MFMA 32x32x16 layout check: max err %.3g (%s)\n", err, err < 1e-3 ? "OK" : "MISMATCH");
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("device %s CUs %d\n", p.gcnArchName, p.multiProcessorCount);
  return 0;
}
