// Development harness: times sbr_gemm_f32 / sbr_gemm_tn_f32 of whichever gemm_f32 variant it is linked with (hipEvents).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <vector>
extern "C" int sbr_gemm_f32(int mode, const float* A, long lda, const int* a_idx, const float* B, long ldb, const int* b_idx,
                            const float* bias, float* C, long ldc, const int* c_idx, int M, int N, int K, int act, int atomic, void* stream);
extern "C" long sbr_gemm_tn_f32_workspace(int M, int N, int K);
extern "C" int sbr_gemm_tn_f32(const float* A, long lda, const int* a_idx, const float* B, long ldb, const int* b_idx,
                               float* C, long ldc, int M, int N, int K, void* workspace, long workspace_bytes, void* stream);
void sbr_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); }

static float* dev_rand(size_t n, unsigned seed) {
  std::vector<float> h(n);
  unsigned s = seed;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xFFFF) / 65536.f - 0.5f; }
  float* d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); return d;
}

template <class F> static float time_ms(F f, int reps = 20) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

int main(int argc, char** argv) {
  const char* tag = argc > 1 ? argv[1] : "base";
  struct Case { const char* name; int mode, M, N, K, gather; };
  Case cases[] = {{"NT mlp 90112x128x128", 0, 90112, 128, 128, 0}, {"NT proj 45056x128x768 g", 0, 45056, 128, 768, 1},
                  {"NN dx 90112x128x128", 1, 90112, 128, 128, 0}, {"NT score 8192x50000x128", 0, 8192, 50000, 128, 0},
                  {"NT sq 4096^3", 0, 4096, 4096, 4096, 0}};
  for (auto& c : cases) {
    const int rowsA = c.gather ? 50000 : c.M;
    float* A = dev_rand((size_t)rowsA * c.K, 1);
    float* B = dev_rand((size_t)(c.mode == 1 ? c.K : c.N) * (c.mode == 1 ? c.N : c.K), 2);
    float* C; hipMalloc(&C, (size_t)c.M * c.N * 4);
    int* idx = nullptr;
    if (c.gather) {
      std::vector<int> h(c.M); unsigned s = 7;
      for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 8) % rowsA; }
      hipMalloc(&idx, c.M * 4); hipMemcpy(idx, h.data(), c.M * 4, hipMemcpyHostToDevice);
    }
    const long ldb = c.mode == 1 ? c.N : c.K;
    float ms = time_ms([&] { sbr_gemm_f32(c.mode, A, c.K, idx, B, ldb, nullptr, nullptr, C, c.N, nullptr, c.M, c.N, c.K, 0, 0, nullptr); });
    printf("%-8s %-28s %9.1f us %8.2f TFLOP/s\n", tag, c.name, ms * 1e3, 2.0 * c.M * c.N * c.K / ms / 1e9);
    if (c.mode == 0 && c.N <= 1024) {
      float* bias = dev_rand(c.N, 3);
      ms = time_ms([&] { sbr_gemm_f32(c.mode, A, c.K, idx, B, ldb, nullptr, bias, C, c.N, nullptr, c.M, c.N, c.K, 1, 0, nullptr); });
      printf("%-8s %-28s %9.1f us %8.2f TFLOP/s  (+bias, relu)\n", tag, c.name, ms * 1e3, 2.0 * c.M * c.N * c.K / ms / 1e9);
      hipFree(bias);
    }
    hipFree(A); hipFree(B); hipFree(C); if (idx) hipFree(idx);
  }
  struct TCase { const char* name; int M, N, K, gather; };
  TCase tcases[] = {{"TN dW 128x128 K=90112", 128, 128, 90112, 0}, {"TN dWproj 128x768 K=45056 g", 128, 768, 45056, 1}};
  for (auto& c : tcases) {
    const int rowsB = c.gather ? 50000 : c.K;
    float* A = dev_rand((size_t)c.K * c.M, 1);
    float* B = dev_rand((size_t)rowsB * c.N, 2);
    float* C; hipMalloc(&C, (size_t)c.M * c.N * 4);
    long wsb = sbr_gemm_tn_f32_workspace(c.M, c.N, c.K);
    void* ws; hipMalloc(&ws, wsb);
    int* idx = nullptr;
    if (c.gather) {
      std::vector<int> h(c.K); unsigned s = 7;
      for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 8) % rowsB; }
      hipMalloc(&idx, c.K * 4); hipMemcpy(idx, h.data(), c.K * 4, hipMemcpyHostToDevice);
    }
    float ms = time_ms([&] { sbr_gemm_tn_f32(A, c.M, nullptr, B, c.N, idx, C, c.N, c.M, c.N, c.K, ws, wsb, nullptr); });
    printf("%-8s %-28s %9.1f us %8.2f TFLOP/s\n", tag, c.name, ms * 1e3, 2.0 * c.M * c.N * c.K / ms / 1e9);
    hipFree(A); hipFree(B); hipFree(C); hipFree(ws); if (idx) hipFree(idx);
  }
  return 0;
}
